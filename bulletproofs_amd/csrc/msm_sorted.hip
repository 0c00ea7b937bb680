// Large-N Pippenger MSM for secp256k1 on CDNA4 (gfx950): the raw-MSM configuration of the reference's
// fold idiom (/root/reference/src/proofs/range_proof.rs:770-774 at 2^16..2^22 terms).
//
// The LDS path of msm_kernels.hip gives every (slice, window) workgroup its own bucket set, so the
// bucket-reduction additions grow with the slice count and one workgroup per CU leaves the SIMDs at
// one wave.  Here each window has ONE bucket set in HBM and the work is made uniform per lane:
//
//   k_srt_hist     (chunk, window) workgroups: LDS histogram of |digit|, written per chunk (no global atomics)
//   k_srt_colscan  exclusive scan over the chunks of every bucket count, bucket sizes out
//   k_srt_bstart   per window: exclusive scan of the bucket sizes -> bucket start offsets
//   k_srt_scatter  (chunk, window): LDS cursors; writes the window's points sorted by bucket
//                  (8-byte entry = virtual point index | sign bit, bucket id)
//   k_srt_acc      one lane per SEGMENT of L consecutive sorted entries (not per bucket): every lane
//                  performs exactly L mixed additions whatever the bucket sizes are.  A run that starts
//                  inside the segment is the beginning of its bucket and is written to Bk[bucket]; the
//                  run at the segment start goes to Pseg[segment].
//   k_srt_finalize one lane per bucket: Bk[b] (+)= the Pseg of every segment boundary inside the bucket;
//                  buckets spanning many segments are queued and summed by one workgroup each (k_srt_finalize_big)
//   k_srt_rowcol   bucket index b = hi*LO + lo: row sums R_hi and column sums C_lo, one wave each
//   k_srt_window   sum_b (b+1) B_b = sum_lo lo*C_lo + sum_hi (LO*hi + 1) R_hi by an in-LDS suffix scan
//
// No atomics on group elements; the order of additions inside a bucket depends on the LDS cursor
// race in k_srt_scatter, which cannot change the (canonical, affine) result.
#include <hip/hip_runtime.h>

#include "ec.h"
#include "kernels.h"

namespace bp {

#define SRT_T 1024

__global__ __launch_bounds__(SRT_T) void k_srt_hist(const int16_t* __restrict__ digits, size_t n2, size_t npad, int CH, int NB,
                                                    uint32_t* __restrict__ cnt) {
    extern __shared__ uint32_t srt_h[];
    const int tid = threadIdx.x, chunk = blockIdx.x, w = blockIdx.y, nchunks = gridDim.x;
    for (int b = tid; b < NB; b += SRT_T) srt_h[b] = 0;
    __syncthreads();
    const size_t base = (size_t)chunk * CH;
    const int len = base >= n2 ? 0 : (int)((n2 - base) < (size_t)CH ? (n2 - base) : (size_t)CH);
    const int16_t* dg = digits + (size_t)w * npad + base;
    for (int i = tid; i < len; i += SRT_T) {
        int d = dg[i];
        if (d != 0) atomicAdd(&srt_h[(d < 0 ? -d : d) - 1], 1u);
    }
    __syncthreads();
    uint32_t* o = cnt + ((size_t)w * nchunks + chunk) * NB;
    for (int b = tid; b < NB; b += SRT_T) o[b] = srt_h[b];
}

// cnt[w][chunk][b] -> exclusive prefix over chunks (in place); sz[w][b] = bucket size
__global__ void k_srt_colscan(uint32_t* __restrict__ cnt, int nchunks, int NB, uint32_t* __restrict__ sz) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x, w = blockIdx.y;
    if (b >= NB) return;
    uint32_t tot = 0;
    for (int ch = 0; ch < nchunks; ch++) {
        uint32_t* p = cnt + ((size_t)w * nchunks + ch) * NB + b;
        uint32_t v = *p;
        *p = tot;
        tot += v;
    }
    sz[(size_t)w * (NB + 1) + b] = tot;
}

// in place: sizes -> exclusive starts, entry NB = number of non-zero digits of the window
__global__ __launch_bounds__(SRT_T) void k_srt_bstart(uint32_t* __restrict__ bstart, int NB) {
    __shared__ uint32_t ts[SRT_T];
    const int tid = threadIdx.x, w = blockIdx.x;
    uint32_t* a = bstart + (size_t)w * (NB + 1);
    const int per = (NB + SRT_T - 1) / SRT_T;
    int lo = tid * per, hi = lo + per;
    if (lo > NB) lo = NB;
    if (hi > NB) hi = NB;
    uint32_t s = 0;
    for (int b = lo; b < hi; b++) s += a[b];
    ts[tid] = s;
    __syncthreads();
    for (int off = 1; off < SRT_T; off <<= 1) {
        uint32_t v = ts[tid];
        uint32_t u = tid >= off ? ts[tid - off] : 0;
        __syncthreads();
        ts[tid] = v + u;
        __syncthreads();
    }
    uint32_t run = ts[tid] - s;
    for (int b = lo; b < hi; b++) {
        uint32_t v = a[b];
        a[b] = run;
        run += v;
    }
    if (tid == SRT_T - 1) a[NB] = ts[SRT_T - 1];
}

__global__ __launch_bounds__(SRT_T) void k_srt_scatter(const int16_t* __restrict__ digits, size_t n2, size_t npad, int CH, int NB,
                                                       const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ bstart,
                                                       uint2* __restrict__ ent) {
    extern __shared__ uint32_t srt_h[];
    const int tid = threadIdx.x, chunk = blockIdx.x, w = blockIdx.y, nchunks = gridDim.x;
    const uint32_t* o = cnt + ((size_t)w * nchunks + chunk) * NB;
    const uint32_t* bs = bstart + (size_t)w * (NB + 1);
    for (int b = tid; b < NB; b += SRT_T) srt_h[b] = bs[b] + o[b];
    __syncthreads();
    const size_t base = (size_t)chunk * CH;
    const int len = base >= n2 ? 0 : (int)((n2 - base) < (size_t)CH ? (n2 - base) : (size_t)CH);
    const int16_t* dg = digits + (size_t)w * npad + base;
    uint2* E = ent + (size_t)w * npad;
    for (int i = tid; i < len; i += SRT_T) {
        int d = dg[i];
        if (d != 0) {
            int b = (d < 0 ? -d : d) - 1;
            uint32_t pos = atomicAdd(&srt_h[b], 1u);
            E[pos] = make_uint2((uint32_t)(base + i) | (d < 0 ? 0x80000000u : 0u), (uint32_t)b);  // one 8-byte store per entry
        }
    }
}

__global__ __launch_bounds__(256, 2) void k_srt_acc(const affine_mem* __restrict__ pts, const affine_mem* __restrict__ lam, size_t n1,
                                                    const uint2* __restrict__ ent, const uint32_t* __restrict__ bstart, size_t npad, int NB, int L, int nseg,
                                                    int w0, int nw, xyzz* __restrict__ Bk, xyzz* __restrict__ Pseg) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int w = w0 + (int)(g / nseg), t = (int)(g % nseg);
    if (w >= w0 + nw) return;
    const uint32_t Nw = bstart[(size_t)w * (NB + 1) + NB];
    const uint32_t j0 = (uint32_t)t * L;
    if (j0 >= Nw) return;
    const uint32_t j1 = (j0 + L < Nw) ? j0 + L : Nw;
    const uint2* E = ent + (size_t)w * npad;
    xyzz acc = xyzz_inf();
    int cur = (int)E[j0].y;
    bool atstart = true;
    for (uint32_t j = j0; j < j1; j++) {
        const uint2 ev = E[j];
        int b = (int)ev.y;
        if (b != cur) {
            if (atstart)
                Pseg[(size_t)w * nseg + t] = acc;
            else
                Bk[(size_t)w * NB + cur] = acc;
            acc = xyzz_inf();
            cur = b;
            atstart = false;
        }
        uint32_t e = ev.x;
        size_t v = e & 0x7fffffffu;
        affine p = affine_load(v < n1 ? pts[v] : lam[v - n1]);
        if (e >> 31) p.y = fp_neg(p.y);
        xyzz_madd(acc, p);
    }
    if (atstart)
        Pseg[(size_t)w * nseg + t] = acc;
    else
        Bk[(size_t)w * NB + cur] = acc;
}

// Buckets with at most SRT_BIG segment partials are summed by their own lane; larger ones (skewed
// digit distributions: the short top window, +-1 / small scalars) are queued for k_srt_finalize_big
// so that no lane runs a long serial chain of additions.
#define SRT_BIG 16
__global__ __launch_bounds__(256, 2) void k_srt_finalize(const uint32_t* __restrict__ bstart, int NB, int L, int nseg, int w0, int nw,
                                                         xyzz* __restrict__ Bk, const xyzz* __restrict__ Pseg, uint32_t* __restrict__ queue) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int w = w0 + (int)(g / NB), b = (int)(g % NB);
    if (w >= w0 + nw) return;
    const uint32_t s = bstart[(size_t)w * (NB + 1) + b], e = bstart[(size_t)w * (NB + 1) + b + 1];
    xyzz r = xyzz_inf();
    if (e > s) {
        const uint32_t t0 = (s + L - 1) / L;
        const uint32_t t1 = (e + L - 1) / L;  // segments t0 .. t1-1 start inside [s, e)
        if (t1 > t0 + SRT_BIG) {
            uint32_t q = atomicAdd(&queue[0], 1u);
            queue[1 + q] = (uint32_t)w * (uint32_t)NB + (uint32_t)b;
            return;
        }
        if (s % L) r = Bk[(size_t)w * NB + b];
        for (uint32_t t = t0; t < t1; t++) {
            xyzz o = Pseg[(size_t)w * nseg + t];
            xyzz_add(r, o);
        }
    }
    Bk[(size_t)w * NB + b] = r;
}

// one 256-lane workgroup per queued bucket: lanes stride over the bucket's segment partials, LDS tree,
// lane 0 stores (a bucket holding every entry of a window, 2^21 / L partials, costs 256 + 8 addition steps)
__global__ __launch_bounds__(256) void k_srt_finalize_big(const uint32_t* __restrict__ bstart, int NB, int L, int nseg, xyzz* __restrict__ Bk,
                                                          const xyzz* __restrict__ Pseg, const uint32_t* __restrict__ queue) {
    __shared__ xyzz red[256];
    const int lane = threadIdx.x;
    const uint32_t nq = queue[0];
    for (uint32_t q = blockIdx.x; q < nq; q += gridDim.x) {
        const uint32_t g = queue[1 + q];
        const int w = (int)(g / NB), b = (int)(g % NB);
        const uint32_t s = bstart[(size_t)w * (NB + 1) + b], e = bstart[(size_t)w * (NB + 1) + b + 1];
        const uint32_t t0 = (s + L - 1) / L, t1 = (e + L - 1) / L;
        xyzz r = xyzz_inf();
        if (lane == 0 && (s % L)) r = Bk[(size_t)w * NB + b];
        for (uint32_t t = t0 + lane; t < t1; t += 256) {
            xyzz o = Pseg[(size_t)w * nseg + t];
            xyzz_add(r, o);
        }
        red[lane] = r;
        __syncthreads();
        for (int off = 128; off >= 1; off >>= 1) {
            if (lane < off) {
                xyzz v = red[lane];
                xyzz u = red[lane + off];
                xyzz_add(v, u);
                red[lane] = v;
            }
            __syncthreads();
        }
        if (lane == 0) Bk[(size_t)w * NB + b] = red[0];
        __syncthreads();
    }
}

// one wave per output: o < LO -> C_o = sum_hi B[hi*LO + o];  o >= LO -> R_(o-LO) = sum_lo B[(o-LO)*LO + lo]
__global__ __launch_bounds__(256) void k_srt_rowcol(const xyzz* __restrict__ Bk, int NB, int lo_bits, int w0, xyzz* __restrict__ RC) {
    __shared__ xyzz red[256];
    const int tid = threadIdx.x, lane = tid & 63, w = w0 + blockIdx.y;
    const int LO = 1 << lo_bits, HI = NB >> lo_bits, nout = LO + HI;
    const int o = blockIdx.x * 4 + (tid >> 6);
    const xyzz* B = Bk + (size_t)w * NB;
    xyzz acc = xyzz_inf();
    if (o < LO) {
        for (int hi = lane; hi < HI; hi += 64) {
            xyzz v = B[(size_t)hi * LO + o];
            xyzz_add(acc, v);
        }
    } else if (o < nout) {
        for (int lo = lane; lo < LO; lo += 64) {
            xyzz v = B[(size_t)(o - LO) * LO + lo];
            xyzz_add(acc, v);
        }
    }
    red[tid] = acc;
    __syncthreads();
    for (int off = 32; off >= 1; off >>= 1) {
        if (lane < off) {
            xyzz v = red[tid];
            xyzz u = red[tid + off];
            xyzz_add(v, u);
            red[tid] = v;
        }
        __syncthreads();
    }
    if (lane == 0 && o < nout) RC[(size_t)w * nout + o] = red[tid];
}

// wsum[w] = sum_{lo >= 1} suffixC[lo] + 2^lo_bits * sum_{hi >= 1} suffixR[hi] + suffixR[0];  LO + HI <= 256
__global__ __launch_bounds__(256) void k_srt_window(const xyzz* __restrict__ RC, int NB, int lo_bits, int w0, xyzz* __restrict__ wsum) {
    __shared__ xyzz red[256];
    const int tid = threadIdx.x, w = w0 + blockIdx.x;
    const int LO = 1 << lo_bits, HI = NB >> lo_bits, nout = LO + HI;
    xyzz v = tid < nout ? RC[(size_t)w * nout + tid] : xyzz_inf();
    red[tid] = v;
    __syncthreads();
    const int lim = tid < LO ? LO : nout;
    const int span = LO > HI ? LO : HI;
    for (int off = 1; off < span; off <<= 1) {
        bool has = tid < nout && tid + off < lim;
        xyzz u;
        if (has) u = red[tid + off];
        __syncthreads();
        if (has) {
            xyzz_add(v, u);
            red[tid] = v;
        }
        __syncthreads();
    }
    xyzz val = xyzz_inf();
    if (tid < LO) {
        if (tid >= 1) val = v;
    } else if (tid < nout) {
        val = (tid == LO) ? v : xyzz_dbl_n(v, lo_bits);
    }
    red[tid] = val;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (tid < off) {
            xyzz a = red[tid];
            xyzz u = red[tid + off];
            xyzz_add(a, u);
            red[tid] = a;
        }
        __syncthreads();
    }
    if (tid == 0) wsum[w] = red[0];
}

// wsum[w_lo] = sum_{w_lo <= w < w_hi} 2^(c (w - w_lo)) wsum[w]: Horner over the top windows, one lane,
// so that this serial chain can run on a second stream while the low windows are still accumulating.
__global__ void k_srt_horner_part(xyzz* __restrict__ wsum, int w_lo, int w_hi, int c) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    xyzz acc = wsum[w_hi - 1];
    for (int w = w_hi - 2; w >= w_lo; w--) {
        acc = xyzz_dbl_n(acc, c);
        xyzz o = wsum[w];
        xyzz_add(acc, o);
    }
    wsum[w_lo] = acc;
}

// bucket accumulation + reduction of the windows [w0, w0 + nw) on stream st
static void srt_acc_windows(const affine_mem* pts, const affine_mem* lam, size_t n1, const SortedPlan& pl, const SortedBufs& b, int w0, int nw,
                            hipStream_t st) {
    size_t nthr = (size_t)nw * pl.nseg;
    hipLaunchKernelGGL(k_srt_acc, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, pts, lam, n1, b.ent, b.bstart, pl.npad, pl.NB, pl.L,
                       pl.nseg, w0, nw, b.Bk, b.Pseg);
}
static hipError_t srt_reduce_windows(const SortedPlan& pl, const SortedBufs& b, int w0, int nw, uint32_t* queue, xyzz* wsum, hipStream_t st) {
    const int NB = pl.NB;
    hipError_t e = hipMemsetAsync(queue, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    size_t nthr = (size_t)nw * NB;
    hipLaunchKernelGGL(k_srt_finalize, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, b.bstart, NB, pl.L, pl.nseg, w0, nw, b.Bk, b.Pseg,
                       queue);
    hipLaunchKernelGGL(k_srt_finalize_big, dim3(2048), dim3(256), 0, st, b.bstart, NB, pl.L, pl.nseg, b.Bk, b.Pseg, queue);
    const int nout = (1 << pl.lo_bits) + (NB >> pl.lo_bits);
    hipLaunchKernelGGL(k_srt_rowcol, dim3((unsigned)((nout + 3) / 4), (unsigned)nw), dim3(256), 0, st, b.Bk, NB, pl.lo_bits, w0, b.RC);
    hipLaunchKernelGGL(k_srt_window, dim3((unsigned)nw), dim3(256), 0, st, b.RC, NB, pl.lo_bits, w0, wsum);
    return hipGetLastError();
}

// Window sums of the sorted path.  With a second stream (st2 + events) and split_w > 0, the windows
// [split_w, W) are accumulated first; their latency-bound tail (finalize, row/column sums, the window
// scan and the Horner chain over them) then runs on st2 while the windows [0, split_w) accumulate on st.
// On return (stream order of st) wsum[0 .. *W_eff) are the window sums to combine:
// *W_eff = W without the split, split_w + 1 with it (wsum[split_w] = the combined top part).
hipError_t launch_msm_sorted(const affine_mem* pts, const affine_mem* lam, size_t n1, const int16_t* digits, size_t n2, const SortedPlan& pl,
                             const SortedBufs& b, xyzz* wsum, hipStream_t st, hipStream_t st2, hipEvent_t ev_fork, hipEvent_t ev_join, int split_w,
                             int* W_eff) {
    const int NB = pl.NB, W = pl.W;
    const size_t lds = (size_t)NB * 4;
    dim3 gcw((unsigned)pl.nchunks, (unsigned)W);
    hipLaunchKernelGGL(k_srt_hist, gcw, dim3(SRT_T), lds, st, digits, n2, pl.npad, pl.CH, NB, b.cnt);
    hipLaunchKernelGGL(k_srt_colscan, dim3((unsigned)((NB + 255) / 256), (unsigned)W), dim3(256), 0, st, b.cnt, pl.nchunks, NB, b.bstart);
    hipLaunchKernelGGL(k_srt_bstart, dim3((unsigned)W), dim3(SRT_T), 0, st, b.bstart, NB);
    hipLaunchKernelGGL(k_srt_scatter, gcw, dim3(SRT_T), lds, st, digits, n2, pl.npad, pl.CH, NB, b.cnt, b.bstart, b.ent);
    hipError_t e;
    if (!st2 || split_w <= 0 || split_w >= W - 1) {
        srt_acc_windows(pts, lam, n1, pl, b, 0, W, st);
        *W_eff = W;
        return srt_reduce_windows(pl, b, 0, W, b.queue, wsum, st);
    }
    uint32_t* queue2 = b.queue + ((size_t)W * NB + 1);
    srt_acc_windows(pts, lam, n1, pl, b, split_w, W - split_w, st);
    if ((e = hipEventRecord(ev_fork, st)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(st2, ev_fork, 0)) != hipSuccess) return e;
    if ((e = srt_reduce_windows(pl, b, split_w, W - split_w, queue2, wsum, st2)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_srt_horner_part, dim3(1), dim3(64), 0, st2, wsum, split_w, W, pl.c);
    if ((e = hipEventRecord(ev_join, st2)) != hipSuccess) return e;
    srt_acc_windows(pts, lam, n1, pl, b, 0, split_w, st);
    if ((e = srt_reduce_windows(pl, b, 0, split_w, b.queue, wsum, st)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(st, ev_join, 0)) != hipSuccess) return e;
    *W_eff = split_w + 1;
    return hipGetLastError();
}

}  // namespace bp
