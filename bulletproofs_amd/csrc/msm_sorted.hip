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
//   k_srt_rowcol2  bucket index b = hi*LO + lo: row sums R_hi and column sums C_lo, one workgroup each, taken
//                  directly from Bk and Pseg (a sum over buckets needs their partials in no particular order,
//                  so there is no per-bucket finalize pass)
//   k_srt_window   sum_b (b+1) B_b = sum_lo lo*C_lo + sum_hi (LO*hi + 1) R_hi by an in-LDS suffix scan
//
// No atomics on group elements; the order of additions inside a bucket depends on the LDS cursor
// race in k_srt_scatter, which cannot change the (canonical, affine) result.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "ec.h"
#include "kernels.h"

namespace bp {

// 512-lane workgroups (two waves per SIMD, <= 64 VGPRs): a sort workgroup of a later window group fits beside the two
// accumulation waves per SIMD of the group before it (launch_msm_sorted)
#define SRT_T 512

// Digits are read eight at a time (one 16-byte load per lane and step: a chunk of 65536 digits is 8 dependent
// load -> LDS-atomic steps per lane instead of 64); the columns [n2, npad) hold zero digits, so whole chunks are read.
__device__ __forceinline__ int srt_digit(const uint4& q, int h) {
    const uint32_t wd = (h >> 1) == 0 ? q.x : (h >> 1) == 1 ? q.y : (h >> 1) == 2 ? q.z : q.w;
    return (int)(int16_t)((h & 1) ? (wd >> 16) : (wd & 0xffffu));
}

__global__ __launch_bounds__(SRT_T) void k_srt_hist(const int16_t* __restrict__ digits, size_t n2, size_t npad, int CH, int NB, int w0,
                                                    uint32_t* __restrict__ cnt) {
    extern __shared__ uint32_t srt_h[];
    const int tid = threadIdx.x, chunk = blockIdx.x, w = w0 + blockIdx.y, nchunks = gridDim.x;
    for (int b = tid; b < NB; b += SRT_T) srt_h[b] = 0;
    __syncthreads();
    const size_t base = (size_t)chunk * CH;
    const uint4* dg = reinterpret_cast<const uint4*>(digits + (size_t)w * npad + base);
    if (base < n2) {
        for (int i = tid; i < CH / 8; i += SRT_T) {
            const uint4 q = dg[i];
#pragma unroll
            for (int h = 0; h < 8; h++) {
                const int d = srt_digit(q, h);
                if (d != 0) atomicAdd(&srt_h[(d < 0 ? -d : d) - 1], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t* o = cnt + ((size_t)w * nchunks + chunk) * NB;
    for (int b = tid; b < NB; b += SRT_T) o[b] = srt_h[b];
}

// cnt[w][chunk][b] -> exclusive prefix over chunks (in place); sz[w][b] = bucket size
__global__ void k_srt_colscan(uint32_t* __restrict__ cnt, int nchunks, int NB, int w0, uint32_t* __restrict__ sz) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x, w = w0 + blockIdx.y;
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
__global__ __launch_bounds__(SRT_T) void k_srt_bstart(uint32_t* __restrict__ bstart, int NB, int w0) {
    __shared__ uint32_t ts[SRT_T];
    const int tid = threadIdx.x, w = w0 + blockIdx.x;
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

__global__ __launch_bounds__(SRT_T) void k_srt_scatter(const int16_t* __restrict__ digits, size_t n2, size_t npad, int CH, int NB, int w0,
                                                       const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ bstart,
                                                       uint2* __restrict__ ent) {
    extern __shared__ uint32_t srt_h[];
    const int tid = threadIdx.x, chunk = blockIdx.x, w = w0 + blockIdx.y, nchunks = gridDim.x;
    const uint32_t* o = cnt + ((size_t)w * nchunks + chunk) * NB;
    const uint32_t* bs = bstart + (size_t)w * (NB + 1);
    for (int b = tid; b < NB; b += SRT_T) srt_h[b] = bs[b] + o[b];
    __syncthreads();
    const size_t base = (size_t)chunk * CH;
    if (base >= n2) return;
    const uint4* dg = reinterpret_cast<const uint4*>(digits + (size_t)w * npad + base);
    uint2* E = ent + (size_t)w * npad;
    for (int i = tid; i < CH / 8; i += SRT_T) {
        const uint4 q = dg[i];
#pragma unroll
        for (int h = 0; h < 8; h++) {
            const int d = srt_digit(q, h);
            if (d != 0) {
                const int b = (d < 0 ? -d : d) - 1;
                const uint32_t pos = atomicAdd(&srt_h[b], 1u);
                E[pos] = make_uint2((uint32_t)(base + 8 * i + h) | (d < 0 ? 0x80000000u : 0u), (uint32_t)b);  // one 8-byte store per entry
            }
        }
    }
}

// The loop is software-pipelined so that a wave does not need its neighbours to hide the gathers (a group of few
// windows runs at one or two waves per SIMD): the point of entry j + 1 is requested before the addition of entry j
// starts.  Entries are read one 64-byte line (8 entries, four 16-byte loads) at a time into an entry-major LDS slot
// (conflict-free 8-byte reads); j0 is a multiple of 8 and the window's array 64-byte aligned.  With one 8-byte global
// load per iteration the line was fetched again for every entry (the point gathers sweep the caches in between):
// TCC_EA read requests 27.5 M per launch against 10.6 M algorithmic.
__global__ __launch_bounds__(256, 2) void k_srt_acc(const affine_mem* __restrict__ pts, const affine_mem* __restrict__ lam, size_t n1,
                                                    const uint2* __restrict__ ent, const uint32_t* __restrict__ bstart, size_t npad, int NB, int L, int nseg,
                                                    int pstride, int w0, int nw, xyzz* __restrict__ Bk, xyzz* __restrict__ Pseg) {
    __shared__ uint2 line[8][256];
    const int tid = threadIdx.x;
    const size_t g = (size_t)blockIdx.x * 256 + tid;
    const int w = w0 + (int)(g / nseg), t = (int)(g % nseg);
    if (w >= w0 + nw) return;
    const uint32_t Nw = bstart[(size_t)w * (NB + 1) + NB];
    const uint32_t j0 = (uint32_t)t * L;
    if (j0 >= Nw) return;
    const uint32_t j1 = (j0 + L < Nw) ? j0 + L : Nw;
    const uint2* E = ent + (size_t)w * npad;
    xyzz acc = xyzz_inf();
    bool atstart = true;
    // line of entries j0 .. j0 + 7 (entries past Nw are never used)
    {
        const uint4* q = reinterpret_cast<const uint4*>(E + j0);
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const uint4 v = q[h];
            line[2 * h][tid] = make_uint2(v.x, v.y);
            line[2 * h + 1][tid] = make_uint2(v.z, v.w);
        }
    }
    uint2 en = line[0][tid];
    int cur = (int)en.y;
    affine_mem nxt;
    {
        const size_t v = en.x & 0x7fffffffu;
        nxt = v < n1 ? pts[v] : lam[v - n1];
    }
    for (uint32_t j = j0; j < j1; j++) {
        const uint2 ec = en;
        affine p = affine_load(nxt);
        if (j + 1 < j1) {
            const uint32_t h = (j + 1 - j0) & 7u;
            if (h == 0) {
                const uint4* q = reinterpret_cast<const uint4*>(E + j + 1);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint4 v = q[k];
                    line[2 * k][tid] = make_uint2(v.x, v.y);
                    line[2 * k + 1][tid] = make_uint2(v.z, v.w);
                }
            }
            en = line[h][tid];
            const size_t v = en.x & 0x7fffffffu;
            nxt = v < n1 ? pts[v] : lam[v - n1];
        }
        const int b = (int)ec.y;
        if (b != cur) {
            if (atstart)
                Pseg[(size_t)w * pstride + t] = acc;
            else
                Bk[(size_t)w * NB + cur] = acc;
            acc = xyzz_inf();
            cur = b;
            atstart = false;
        }
        xyzz_madd_s(acc, p, ec.x >> 31);
    }
    if (atstart)
        Pseg[(size_t)w * pstride + t] = acc;
    else
        Bk[(size_t)w * NB + cur] = acc;
}

// Bucket grid of one window.  A window whose entries all lie in the first NBe = NB / 2^j buckets (always the
// short top window of a GLV half-scalar; any window when the scalars are small) is reduced over an NBe-bucket
// grid b = hi * LOe + lo, so that its few, large buckets spread over all rows and columns instead of a handful.
// Both reduction kernels derive the same grid from bstart (uniform loads).
__device__ __forceinline__ void srt_grid(const uint32_t* __restrict__ bs, int NB, int& lo_bits, int& NBe) {
    const uint32_t Nw = bs[NB];
    int nbits = 31 - __builtin_clz((unsigned)NB);
    while (nbits > 1 && bs[1 << (nbits - 1)] == Nw) nbits--;
    NBe = 1 << nbits;
    if (NBe != NB) lo_bits = (nbits + 1) / 2;
}

// wsum[w] = sum_{lo >= 1} suffixC[lo] + 2^lo_bits * sum_{hi >= 1} suffixR[hi] + suffixR[0];  LO + HI <= 256
__global__ __launch_bounds__(256) void k_srt_window(const uint32_t* __restrict__ bstart, const xyzz* __restrict__ RC, int NB, int lo_bits, int w0,
                                                       xyzz* __restrict__ wsum) {
    __shared__ xyzz red[256];
    __builtin_amdgcn_s_setprio(3);  // latency-bound: issue ahead of the co-resident accumulation waves
    const int tid = threadIdx.x, w = w0 + blockIdx.x;
    const int stride = (1 << lo_bits) + (NB >> lo_bits);
    int NBe;
    srt_grid(bstart + (size_t)w * (NB + 1), NB, lo_bits, NBe);
    const int LO = 1 << lo_bits, HI = NBe >> lo_bits, nout = LO + HI;
    xyzz v = tid < nout ? RC[(size_t)w * stride + tid] : xyzz_inf();
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

// wsum[w_lo] = sum_{w_lo <= w < w_hi} 2^(c (w - w_lo)) wsum[w]  (+ wsum[vidx] when vidx >= 0: the combination of the
// window groups above, already multiplied by 2^(c (w_hi - w_lo)) by k_srt_dbl): Horner over one group's windows, one
// lane, so that this serial chain runs on a side stream while the lower windows are still accumulating.
__global__ __launch_bounds__(64) void k_srt_horner_part(xyzz* __restrict__ wsum, int w_lo, int w_hi, int c, int vidx) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    __builtin_amdgcn_s_setprio(3);
    xyzz acc = wsum[w_hi - 1];
    for (int w = w_hi - 2; w >= w_lo; w--) {
        acc = xyzz_dbl_n(acc, c);
        xyzz o = wsum[w];
        xyzz_add(acc, o);
    }
    if (vidx >= 0) {
        xyzz o = wsum[vidx];
        xyzz_add(acc, o);
    }
    wsum[w_lo] = acc;
}
// wsum[idx] *= 2^k: the doublings the finished groups' combination needs before the next group's windows can be added
// do not wait for that group's bucket reduction
__global__ __launch_bounds__(64) void k_srt_dbl(xyzz* __restrict__ wsum, int idx, int k) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    __builtin_amdgcn_s_setprio(3);
    wsum[idx] = xyzz_dbl_n(wsum[idx], k);
}

// Row and column sums straight from k_srt_acc's outputs (no per-bucket finalize pass): the sum over a set of
// buckets only needs every partial that belongs to one of them - Bk[b] (the run that starts inside a segment)
// and the Pseg[t] of every segment that starts inside the bucket - in any order.  One 256-lane workgroup per
// output: o < LO -> C_o = sum_hi bucket[hi*LO + o];  o >= LO -> R_(o-LO) = sum_lo bucket[(o-LO)*LO + lo].
// A row's segments are one contiguous range; a column gives each of its HI buckets T/HI lanes (at least one), and buckets
// with many segments (the short top window, skewed scalars) are summed by the whole workgroup afterwards.
// Depth: ~2-3 additions per lane + an 8-level LDS tree, whatever the bucket sizes are.
// LDS: blockDim.x XYZZ slots, passed as dynamic shared memory (a 64-lane launch then takes 9 KB, not the 36 KB of 256 lanes:
// with a fixed red[256] only four workgroups fitted a CU and the 64-lane form ran at one wave per SIMD).
__global__ __launch_bounds__(256) void k_srt_rowcol2(const uint32_t* __restrict__ bstart, const xyzz* __restrict__ Bk,
                                                        const xyzz* __restrict__ Pseg, int NB, int L, int pstride, int lo_bits, int w0,
                                                        xyzz* __restrict__ RC) {
    extern __shared__ __align__(16) unsigned char srt_rc_lds[];
    xyzz* red = reinterpret_cast<xyzz*>(srt_rc_lds);
    __builtin_amdgcn_s_setprio(3);  // latency-bound: issue ahead of the co-resident accumulation waves
    __shared__ uint32_t big_t0[256], big_t1[256];
    __shared__ uint32_t nbig;
    const int tid = threadIdx.x, T = blockDim.x, w = w0 + blockIdx.y, o = blockIdx.x;  // T = 64, 128 or 256 lanes per output
    const int stride = (1 << lo_bits) + (NB >> lo_bits);
    const uint32_t* bs = bstart + (size_t)w * (NB + 1);
    int NBe;
    srt_grid(bs, NB, lo_bits, NBe);
    const int LO = 1 << lo_bits, HI = NBe >> lo_bits, nout = LO + HI;
    if (o >= nout) return;
    const xyzz* B = Bk + (size_t)w * NB;
    const xyzz* P = Pseg + (size_t)w * pstride;
    const uint32_t uL = (uint32_t)L;
    xyzz acc = xyzz_inf();
    if (tid == 0) nbig = 0;
    __syncthreads();
    if (o >= LO) {
        const int b0 = (o - LO) * LO;
        for (int i = tid; i < LO; i += T) {
            const uint32_t s = bs[b0 + i], e = bs[b0 + i + 1];
            if (e > s && (s % uL)) {
                xyzz v = B[b0 + i];
                xyzz_add(acc, v);
            }
        }
        const uint32_t t0 = (bs[b0] + uL - 1) / uL, t1 = (bs[b0 + LO] + uL - 1) / uL;
        for (uint32_t t = t0 + tid; t < t1; t += T) {
            xyzz v = P[t];
            xyzz_add(acc, v);
        }
    } else {
        const int Q = HI >= T ? 1 : T / HI;  // lanes per bucket
        for (int h0 = 0; h0 < HI; h0 += T / Q) {
            const int hi = h0 + tid / Q, sub = tid % Q;
            if (hi < HI) {
                const int b = hi * LO + o;
                const uint32_t s = bs[b], e = bs[b + 1];
                if (e > s) {
                    if (sub == 0 && (s % uL)) {
                        xyzz v = B[b];
                        xyzz_add(acc, v);
                    }
                    const uint32_t t0 = (s + uL - 1) / uL, t1 = (e + uL - 1) / uL;
                    if (t1 - t0 <= (uint32_t)(4 * Q < 8 ? 8 : 4 * Q)) {
                        for (uint32_t t = t0 + sub; t < t1; t += Q) {
                            xyzz v = P[t];
                            xyzz_add(acc, v);
                        }
                    } else if (sub == 0) {
                        const uint32_t q = atomicAdd(&nbig, 1u);
                        big_t0[q] = t0;
                        big_t1[q] = t1;
                    }
                }
            }
            __syncthreads();
            const uint32_t nq = nbig;
            for (uint32_t q = 0; q < nq; q++) {
                const uint32_t t1 = big_t1[q];
                for (uint32_t t = big_t0[q] + tid; t < t1; t += T) {
                    xyzz v = P[t];
                    xyzz_add(acc, v);
                }
            }
            __syncthreads();
            if (tid == 0) nbig = 0;
            __syncthreads();
        }
    }
    red[tid] = acc;
    __syncthreads();
    for (int off = T / 2; off >= 1; off >>= 1) {
        if (tid < off) {
            xyzz v = red[tid];
            xyzz u = red[tid + off];
            xyzz_add(v, u);
            red[tid] = v;
        }
        __syncthreads();
    }
    if (tid == 0) RC[(size_t)w * stride + o] = red[0];
}

// counting sort by bucket of the windows [w0, w0 + nw) on stream st
static void srt_sort_windows(const int16_t* digits, size_t n2, const SortedPlan& pl, const SortedBufs& b, int w0, int nw, hipStream_t st) {
    const int NB = pl.NB;
    const size_t lds = (size_t)NB * 4;
    dim3 gcw((unsigned)pl.nchunks, (unsigned)nw);
    hipLaunchKernelGGL(k_srt_hist, gcw, dim3(SRT_T), lds, st, digits, n2, pl.npad, pl.CH, NB, w0, b.cnt);
    hipLaunchKernelGGL(k_srt_colscan, dim3((unsigned)((NB + 255) / 256), (unsigned)nw), dim3(256), 0, st, b.cnt, pl.nchunks, NB, w0, b.bstart);
    hipLaunchKernelGGL(k_srt_bstart, dim3((unsigned)nw), dim3(SRT_T), 0, st, b.bstart, NB, w0);
    hipLaunchKernelGGL(k_srt_scatter, gcw, dim3(SRT_T), lds, st, digits, n2, pl.npad, pl.CH, NB, w0, b.cnt, b.bstart, b.ent);
}
// bucket accumulation of the windows [w0, w0 + nw) on stream st, segments of L entries (nseg per window)
static void srt_acc_windows(const affine_mem* pts, const affine_mem* lam, size_t n1, const SortedPlan& pl, const SortedBufs& b, int w0, int nw, int L,
                            int nseg, hipStream_t st) {
    size_t nthr = (size_t)nw * nseg;
    hipLaunchKernelGGL(k_srt_acc, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, pts, lam, n1, b.ent, b.bstart, pl.npad, pl.NB, L, nseg,
                       pl.nseg, w0, nw, b.Bk, b.Pseg);
}
// bucket reduction of the windows [w0, w0 + nw): wsum[w] = sum_b (b+1) bucket_w[b]
static hipError_t srt_reduce_windows(const SortedPlan& pl, const SortedBufs& b, int w0, int nw, int L, xyzz* wsum, hipStream_t st) {
    const int NB = pl.NB;
    const int nout = (1 << pl.lo_bits) + (NB >> pl.lo_bits);
    // lanes per output: 256 measured best at 2^20 and 2^22 (128 equal, 64 loses 12 %: the longer serial part per lane
    // costs more than the two tree levels it saves; gpurun sweep of BP_MSM_RC_T)
    int T = 256;
    if (const char* e = getenv("BP_MSM_RC_T")) T = atoi(e) == 64 || atoi(e) == 128 || atoi(e) == 256 ? atoi(e) : T;
    hipLaunchKernelGGL(k_srt_rowcol2, dim3((unsigned)nout, (unsigned)nw), dim3((unsigned)T), (size_t)T * sizeof(xyzz), st, b.bstart, b.Bk, b.Pseg, NB, L, pl.nseg,
                       pl.lo_bits, w0, b.RC);
    hipLaunchKernelGGL(k_srt_window, dim3((unsigned)nw), dim3(256), 0, st, b.bstart, b.RC, NB, pl.lo_bits, w0, wsum);
    return hipGetLastError();
}

// Window sums of the sorted path.
// ng <= 1: every window accumulated and reduced on st; wsum[0 .. W) are the window sums, *W_eff = W, *c_eff = c.
// ng >= 2 (needs ss): the windows are processed top-down in groups of gsz[0], gsz[1], ... windows.  The groups
// accumulate back to back on st.  Side streams: st2 sorts the later groups' windows beside the first accumulation and
// then runs the Horner chain (st4 is an alias of st2: the process has four hardware queues and a third side stream
// shared st3's), st3 runs each finished group's bucket reduction.  Horner chain: V <- 2^(c gsz[g]) V is queued as soon
// as V (the combination of the groups above g) exists, the group's own chain sum_w 2^(c (w - lo)) S_w + V follows its
// reduction.  A last group of ONE window is not combined here: wsum[0] = S_0, wsum[1] = the doubled V, *W_eff = 2,
// *c_eff = 0 - the caller's final kernel adds them on its way to the affine result.
// Otherwise on return (stream order of st) wsum[0] is the complete combination and *W_eff = 1.
hipError_t launch_msm_sorted(const affine_mem* pts, const affine_mem* lam, size_t n1, const int16_t* digits, size_t n2, const SortedPlan& pl,
                             const SortedBufs& b, xyzz* wsum, hipStream_t st, const SortedStreams* ss, const int* gsz, int ng, int* W_eff,
                             int* c_eff) {
    const int W = pl.W;
    hipError_t e;
    *c_eff = pl.c;
    if (!ss || ng <= 1) {
        srt_sort_windows(digits, n2, pl, b, 0, W, st);
        srt_acc_windows(pts, lam, n1, pl, b, 0, W, pl.L, pl.nseg, st);
        *W_eff = W;
        return srt_reduce_windows(pl, b, 0, W, pl.L, wsum, st);
    }
    // BP_MSM_LANES=<lanes> (tuning): longer segments per group, so that a group's accumulation launch has about that many
    // lanes (2 * 256 * CUs = two waves per SIMD, leaving wave slots to the side streams).  Measured at 2^20: no gain - the
    // side kernels slow the accumulation through the memory system (sort traffic evicts the points from the Infinity
    // Cache), not through wave slots - so the default keeps the plan's segment length for every group.
    size_t lanes = 0;
    if (const char* e2 = getenv("BP_MSM_LANES")) lanes = (size_t)atol(e2) >= 4096 ? (size_t)atol(e2) : 0;
    // the top group's windows are sorted first, the other groups' on st2, group by group, beside the accumulation
    srt_sort_windows(digits, n2, pl, b, W - gsz[0], gsz[0], st);
    if ((e = hipEventRecord(ss->ev_fork, st)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(ss->st2, ss->ev_fork, 0)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(ss->st4, ss->ev_fork, 0)) != hipSuccess) return e;
    for (int g = 1, top = W - gsz[0]; g < ng; g++) {
        srt_sort_windows(digits, n2, pl, b, top - gsz[g], gsz[g], ss->st2);
        if ((e = hipEventRecord(ss->ev_sorted[g], ss->st2)) != hipSuccess) return e;
        top -= gsz[g];
    }
    int hi = W, vidx = -1;
    *W_eff = 1;
    for (int g = 0; g < ng; g++) {
        const int lo = hi - gsz[g];
        int L = pl.L;
        if (lanes && pl.L >= 32) {
            size_t l = ((size_t)gsz[g] * pl.npad + lanes - 1) / lanes;
            l = (l + 7) & ~(size_t)7;
            if (l > 256) l = 256;
            if ((int)l > L) L = (int)l;
        }
        const int nseg = (int)((pl.npad + L - 1) / L);
        if (g >= 1 && (e = hipStreamWaitEvent(st, ss->ev_sorted[g], 0)) != hipSuccess) return e;
        srt_acc_windows(pts, lam, n1, pl, b, lo, hi - lo, L, nseg, st);
        if ((e = hipEventRecord(ss->ev_acc[g], st)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(ss->st3, ss->ev_acc[g], 0)) != hipSuccess) return e;
        if ((e = srt_reduce_windows(pl, b, lo, hi - lo, L, wsum, ss->st3)) != hipSuccess) return e;
        if ((e = hipEventRecord(ss->ev_red[g], ss->st3)) != hipSuccess) return e;
        if (vidx >= 0) hipLaunchKernelGGL(k_srt_dbl, dim3(1), dim3(64), 0, ss->st4, wsum, vidx, pl.c * gsz[g]);
        if ((e = hipStreamWaitEvent(ss->st4, ss->ev_red[g], 0)) != hipSuccess) return e;
        if (g == ng - 1 && gsz[g] == 1 && vidx == 1) {
            *W_eff = 2;
            *c_eff = 0;
        } else {
            hipLaunchKernelGGL(k_srt_horner_part, dim3(1), dim3(64), 0, ss->st4, wsum, lo, hi, pl.c, vidx);
            vidx = lo;
        }
        hi = lo;
    }
    if ((e = hipEventRecord(ss->ev_join, ss->st4)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(st, ss->ev_join, 0)) != hipSuccess) return e;
    return hipGetLastError();
}

}  // namespace bp

BP_OPCOUNT_READER(bp_opcount_msm_sorted)
