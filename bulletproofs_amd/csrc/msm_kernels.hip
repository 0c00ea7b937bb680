// Pippenger bucket MSM for secp256k1 on CDNA4 (gfx950) — generic (variable-base) path.
//
// Replaces the reference's fold idiom
//     (0..n).map(|i| &points[i] * &Scalar::from(&scalars[i])).fold(init, |acc, x| acc + x)
// (/root/reference/src/proofs/range_proof.rs:116-122, 236-244, 770-774; inner_product.rs:73-115,
// 336-338; range_proof_wip.rs:249-251, 370-372, 598-602; weighted_inner_product.rs:109-154, 500-504)
// with: signed-digit recoding -> per-(slice, window) workgroups that counting-sort their points by
// bucket in LDS and sum each bucket in registers (no atomics on group elements, no global sort)
// -> in-LDS suffix-scan bucket reduction -> per-window slice reduction -> Horner combine.
//
// Data layout in HBM: points as `affine_mem` (canonical 8x32-bit words per coordinate, 64 B, one
// cache line per gather; converted to the 9x29-bit compute form after the load);
// digits as int16 [W][Npad] (window-major so a (slice, window) workgroup streams a contiguous,
// coalesced run); partial sums as XYZZ in compute form (144 B).
#include <hip/hip_runtime.h>

#include "ec.h"
#include "ecmul.h"
#include "ecq.h"
#include "ecwave.h"
#include "fq.h"
#include "kernels.h"

namespace bp {

// ---------------------------------------------------------------------------------------------
// ingest: 64-byte big-endian x||y -> limb form, with the on-curve check curv's Point::from_bytes
// performs.  err: bit 0 set if any point is neither infinity (64 zero bytes) nor on the curve.
__global__ void k_ingest_points(const uint8_t* __restrict__ in_be, affine_mem* __restrict__ out, size_t n, int* err) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(in_be + 64 * i);
    affine_mem m;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        m.x.w[k] = __builtin_bswap32(src[7 - k]);
        m.y.w[k] = __builtin_bswap32(src[15 - k]);
    }
    if (!affine_mem_valid(m, true)) atomicOr(err, 1);
    out[i] = m;
}

// limb form -> 64-byte big-endian (infinity -> 64 zero bytes)
__global__ void k_export_points(const affine_mem* __restrict__ in, uint8_t* __restrict__ out_be, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    affine_mem m = in[i];
    uint32_t* dst = reinterpret_cast<uint32_t*>(out_be + 64 * i);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        dst[7 - k] = __builtin_bswap32(m.x.w[k]);
        dst[15 - k] = __builtin_bswap32(m.y.w[k]);
    }
}

// ---------------------------------------------------------------------------------------------
// scalars (32 B big-endian, any 256-bit value; reduced mod q like `Scalar::from(&BigInt)`) ->
// signed c-bit digits d_w in [-(2^(c-1)-1), 2^(c-1)], W = 256/c + 1 windows, window-major.
__device__ __forceinline__ void recode_signed(const uint32_t k[8], int c, int W, bool negate, int16_t* __restrict__ out, size_t stride) {
    uint32_t carry = 0;
    const uint32_t half = 1u << (c - 1);
    const uint32_t mask = (1u << c) - 1;
    for (int w = 0; w < W; w++) {
        int bit = w * c;
        uint32_t v = 0;
        if (bit < 256) {
            v = fq_bits32(k, bit) & mask;
        }
        v += carry;
        int d;
        if (v > half) {
            d = (int)v - (int)(1u << c);
            carry = 1;
        } else {
            d = (int)v;
            carry = 0;
        }
        out[(size_t)w * stride] = (int16_t)(negate ? -d : d);
    }
}

__global__ void k_scalars_to_digits(const uint8_t* __restrict__ scalars_be, size_t n, size_t npad, int c, int W, int16_t* __restrict__ digits) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npad) return;
    if (i >= n) {
        for (int w = 0; w < W; w++) digits[(size_t)w * npad + i] = 0;
        return;
    }
    const uint32_t* src = reinterpret_cast<const uint32_t*>(scalars_be + 32 * i);
    fq k;
#pragma unroll
    for (int j = 0; j < 8; j++) k.v[j] = __builtin_bswap32(src[7 - j]);
    fq_final_reduce(k.v, 0);
    recode_signed(k.v, c, W, false, digits + i, npad);
}

// ---------------------------------------------------------------------------------------------
// GLV front end of every MSM: k*P = k1*P + k2*lambda(P) with |k1|, |k2| < 2^129 and
// lambda(x, y) = (beta*x, y).  An n-term MSM over 256-bit scalars becomes a 2n-term MSM over
// 129-bit scalars: the same number of bucket additions, but half the windows, so the serial
// doubling chain of the final Horner step (one lane, latency-bound) is 128 instead of 256 long.
// Virtual point v < n is points[v], v >= n is lam[v - n].
__global__ void k_glv_expand(const affine_mem* __restrict__ in, affine_mem* __restrict__ lam, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    affine_mem m = in[i];
    if (!affine_mem_is_inf(m)) m.x = fp_to_words(fp_mul(fp_from_words(m.x), fp_beta()));
    lam[i] = m;
}

// digits [W][npad] over the 2n virtual points (columns [2n, npad) zero); sign of k1 / k2 folded in.
__global__ void k_glv_digits(const uint8_t* __restrict__ scalars_be, size_t n, size_t npad, int c, int W, int16_t* __restrict__ digits) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npad) return;
    if (i >= 2 * n) {
        for (int w = 0; w < W; w++) digits[(size_t)w * npad + i] = 0;
        return;
    }
    if (i >= n) return;
    fq k;
    if ((reinterpret_cast<uintptr_t>(scalars_be) & 15) == 0) {  // the usual case: two 16-byte loads per scalar instead of eight dwords
        const uint4* q = reinterpret_cast<const uint4*>(scalars_be + 32 * i);
        const uint4 hi = q[0], lo = q[1];
        k.v[7] = __builtin_bswap32(hi.x), k.v[6] = __builtin_bswap32(hi.y), k.v[5] = __builtin_bswap32(hi.z), k.v[4] = __builtin_bswap32(hi.w);
        k.v[3] = __builtin_bswap32(lo.x), k.v[2] = __builtin_bswap32(lo.y), k.v[1] = __builtin_bswap32(lo.z), k.v[0] = __builtin_bswap32(lo.w);
    } else {
        const uint8_t* b = scalars_be + 32 * i;
#pragma unroll
        for (int j = 0; j < 8; j++)
            k.v[7 - j] = ((uint32_t)b[4 * j] << 24) | ((uint32_t)b[4 * j + 1] << 16) | ((uint32_t)b[4 * j + 2] << 8) | (uint32_t)b[4 * j + 3];
    }
    fq_final_reduce(k.v, 0);
    fq k1, k2;
    bool n1, n2;
    fq_glv_split(k, k1, n1, k2, n2);
    recode_signed(k1.v, c, W, n1, digits + i, npad);
    recode_signed(k2.v, c, W, n2, digits + n + i, npad);
}

// ---------------------------------------------------------------------------------------------
// One workgroup per (slice of <= M points, window w).
//   1. LDS histogram of |digit| (ds_add), exclusive scan, scatter of 16-bit local indices
//      (bit 15 = negate) -> points of this slice sorted by bucket, in LDS.
//   2. Thread t owns the r = NB/T consecutive buckets [t*r, (t+1)*r): sums each bucket in
//      registers (mixed adds from gathered 64-B points), folding them high-to-low into
//      run = sum B_b and tot = sum (b - t*r + 1) B_b.
//   3. Block result sum_b (b+1) B_b = sum_t tot_t + sum_{u>=1} suffix_u(r * run_t): an in-LDS
//      Hillis-Steele suffix scan followed by a tree reduction (r is a power of two -> doublings).
// LDS: 2*(NB+1) u32 + M u16 + T XYZZ.
template <int T>
__global__ __launch_bounds__(T) void k_msm_slice(const affine_mem* __restrict__ pts, const affine_mem* __restrict__ lam, size_t n1,
                                                 const int16_t* __restrict__ digits, size_t n, size_t npad, int M, int c,
                                                 xyzz* __restrict__ partial /* [W][S] */) {
    extern __shared__ __align__(16) uint8_t smem_raw[];
    const int NB = 1 << (c - 1);
    xyzz* red = reinterpret_cast<xyzz*>(smem_raw);                   // T
    uint32_t* start = reinterpret_cast<uint32_t*>(red + T);          // NB + 1
    uint32_t* cursor = start + (NB + 1);                             // NB + 1
    uint16_t* sorted = reinterpret_cast<uint16_t*>(cursor + (NB + 1));  // M

    const int tid = threadIdx.x;
    const int slice = blockIdx.x, w = blockIdx.y, S = gridDim.x;
    const size_t base = (size_t)slice * M;
    const int len = (int)((n - base) < (size_t)M ? (n - base) : (size_t)M);
    const int16_t* dg = digits + (size_t)w * npad + base;

    for (int b = tid; b <= NB; b += T) cursor[b] = 0;
    __syncthreads();
    for (int i = tid; i < len; i += T) {
        int d = dg[i];
        if (d != 0) atomicAdd(&cursor[(d < 0 ? -d : d) - 1], 1u);
    }
    __syncthreads();
    // exclusive scan of cursor[0..NB) -> start[0..NB]; chunked: each thread scans `per` entries
    {
        const int per = (NB + T - 1) / T;
        int lo = tid * per, hi = lo + per;
        if (hi > NB) hi = NB;
        uint32_t s = 0;
        for (int b = lo; b < hi; b++) s += cursor[b];
        uint32_t* tsum = reinterpret_cast<uint32_t*>(red);  // T u32, reuse
        tsum[tid] = s;
        __syncthreads();
        for (int off = 1; off < T; off <<= 1) {
            uint32_t v = tsum[tid];
            uint32_t a = (tid >= off) ? tsum[tid - off] : 0;
            __syncthreads();
            tsum[tid] = v + a;
            __syncthreads();
        }
        uint32_t run = tsum[tid] - s;  // exclusive prefix of this thread's chunk
        for (int b = lo; b < hi; b++) {
            uint32_t cn = cursor[b];
            start[b] = run;
            cursor[b] = run;
            run += cn;
        }
        if (hi == NB && lo < NB) start[NB] = run;
        __syncthreads();
    }
    for (int i = tid; i < len; i += T) {
        int d = dg[i];
        if (d != 0) {
            uint32_t pos = atomicAdd(&cursor[(d < 0 ? -d : d) - 1], 1u);
            sorted[pos] = (uint16_t)(i | (d < 0 ? 0x8000 : 0));
        }
    }
    __syncthreads();

    // bucket accumulation
    const int Tact = NB < T ? NB : T;
    const int r = NB / Tact;  // power of two
    xyzz run = xyzz_inf(), tot = xyzz_inf();
    if (tid < Tact) {
        for (int b = (tid + 1) * r - 1; b >= tid * r; b--) {
            xyzz B = xyzz_inf();
            uint32_t j0 = start[b], j1 = start[b + 1];
            for (uint32_t j = j0; j < j1; j++) {
                uint32_t e = sorted[j];
                size_t v = base + (e & 0x7fff);
                affine p = affine_load(v < n1 ? pts[v] : lam[v - n1]);
                if (e & 0x8000) p.y = fp_neg(p.y);
                xyzz_madd(B, p);
            }
            xyzz_add(run, B);
            xyzz_add(tot, run);
        }
        // r * run
        for (int k = 1; k < r; k <<= 1) run = xyzz_dbl(run);
    }
    __syncthreads();  // tsum alias of red[] no longer needed
    red[tid] = run;
    __syncthreads();
    // inclusive suffix scan: red[t] = sum_{u >= t} r*run_u
    for (int off = 1; off < T; off <<= 1) {
        xyzz v = red[tid];
        bool has = tid + off < T;
        xyzz o;
        if (has) o = red[tid + off];
        __syncthreads();
        if (has) {
            xyzz_add(v, o);
            red[tid] = v;
        }
        __syncthreads();
    }
    // val_t = tot_t + (t >= 1 ? suffix_t : 0); tree-reduce
    {
        xyzz v = tot;
        if (tid >= 1) {
            xyzz s = red[tid];
            xyzz_add(v, s);
        }
        __syncthreads();
        red[tid] = v;
        __syncthreads();
    }
    for (int off = T / 2; off >= 1; off >>= 1) {
        if (tid < off) {
            xyzz v = red[tid];
            xyzz o = red[tid + off];
            xyzz_add(v, o);
            red[tid] = v;
        }
        __syncthreads();
    }
    if (tid == 0) partial[(size_t)w * S + slice] = red[0];
}

template __global__ void k_msm_slice<64>(const affine_mem*, const affine_mem*, size_t, const int16_t*, size_t, size_t, int, int, xyzz*);
template __global__ void k_msm_slice<128>(const affine_mem*, const affine_mem*, size_t, const int16_t*, size_t, size_t, int, int, xyzz*);
template __global__ void k_msm_slice<256>(const affine_mem*, const affine_mem*, size_t, const int16_t*, size_t, size_t, int, int, xyzz*);

// ---------------------------------------------------------------------------------------------
// wsum[w] = sum over slices of partial[w][s]; one 64-lane workgroup per window.
__global__ __launch_bounds__(64) void k_msm_reduce_slices(const xyzz* __restrict__ partial, int S, xyzz* __restrict__ wsum) {
    const int w = blockIdx.x, tid = threadIdx.x;
    xyzz acc = xyzz_inf();
    for (int s = tid; s < S; s += 64) {
        xyzz o = partial[(size_t)w * S + s];
        xyzz_add(acc, o);
    }
    xyzz_wave_tree<64>(acc, tid);  // DPP / ds_bpermute moves (ecwave.h)
    if (tid == 0) wsum[w] = acc;
}

// wsum[w] = sum over parts g of parts[g][w] (multi-GPU combine: one part per rank); one 64-lane workgroup per window.
__global__ __launch_bounds__(64) void k_msm_reduce_parts(const xyzz* __restrict__ parts, int G, int W, xyzz* __restrict__ wsum) {
    const int w = blockIdx.x, tid = threadIdx.x;
    xyzz acc = xyzz_inf();
    for (int g = tid; g < G; g += 64) {
        xyzz o = parts[(size_t)g * W + w];
        if (!fp_raw_is_zero(o.ZZ)) xyzz_add(acc, o);
    }
    xyzz_wave_tree<64>(acc, tid);  // DPP / ds_bpermute moves (ecwave.h)
    if (tid == 0) wsum[w] = acc;
}

// Tiny MSMs (n <= 64: the reference's two- to ten-term folds such as T1 = t1 G + tau1 H, range_proof.rs:168, or the
// eq. (65) sides, :337-354): one workgroup, four lanes per point (ec_mul_glv_slice: GLV half x window range), one LDS
// tree, the division-step inversion.  No buckets, no window sums: the call is pure latency and the longest lane
// (128 doublings + 16 additions) is the floor for a variable base.
#define TINY_MAX 64
__global__ __launch_bounds__(256) void k_msm_tiny(const affine_mem* __restrict__ pts, const uint8_t* __restrict__ scalars_be, int n,
                                                  const affine_mem* __restrict__ init, xyzz* __restrict__ tabs, affine_mem* __restrict__ out_limbs,
                                                  uint8_t* __restrict__ out_be) {
    __shared__ xyzz red[256];
    const int tid = threadIdx.x, v = tid >> 2, slice = tid & 3;
    xyzz acc = xyzz_inf();
    if (v < n) acc = ec_mul_glv_slice(affine_load(pts[v]), fq_from_be_reduce(scalars_be + 32 * (size_t)v), slice, tabs + (size_t)tid * ECMUL_NT);
    xyzz_group_tree_quad<256>(acc, tid, 0, red);  // one level of single-lane additions, seven of quad-shared ones (ecwave.h)
    if (tid != 0) return;
    xyzz r = acc;
    if (init) xyzz_madd(r, affine_load(*init));
    affine_mem m = affine_store(xyzz_to_affine_var(r));
    if (out_limbs) *out_limbs = m;
    if (out_be) {
        uint32_t* dst = reinterpret_cast<uint32_t*>(out_be);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            dst[7 - k] = __builtin_bswap32(m.x.w[k]);
            dst[15 - k] = __builtin_bswap32(m.y.w[k]);
        }
    }
}
size_t msm_tiny_max() { return TINY_MAX; }
size_t msm_tiny_scratch_bytes() { return (size_t)256 * ECMUL_NT * sizeof(xyzz); }
hipError_t launch_msm_tiny(const affine_mem* pts, const uint8_t* scalars_be, size_t n, const affine_mem* init, xyzz* tabs, affine_mem* out_limbs,
                           uint8_t* out_be, hipStream_t st) {
    hipLaunchKernelGGL(k_msm_tiny, dim3(1), dim3(256), 0, st, pts, scalars_be, (int)n, init, tabs, out_limbs, out_be);
    return hipGetLastError();
}

// result = init + sum_w 2^(c*w) wsum[w]  (Horner, one quad of lanes sharing every point operation, ecq.h: the doubling
// chain is inherently serial), normalised to the canonical affine element and written as 64 big-endian bytes.
// out_raw != nullptr (callers that read the result on the host: bp_msm, bp_msm_device): the XYZZ sum and the input-validation
// flag in ONE buffer - the host does the inversion with the same ec.h code (45 us of a lone lane here, 2 us there) and fetches
// sum and flag with one copy instead of two.
__global__ __launch_bounds__(64) void k_msm_horner(const xyzz* __restrict__ wsum, int W, int c, const affine_mem* __restrict__ init, affine_mem* __restrict__ out_limbs,
                             uint8_t* __restrict__ out_be, xyzz* __restrict__ out_raw, const int* __restrict__ err_in) {
    if (threadIdx.x >= 4 || blockIdx.x != 0) return;
    __builtin_amdgcn_s_setprio(3);
    const int q = threadIdx.x;
    xyzz acc = wsum[W - 1];
    for (int w = W - 2; w >= 0; w--) {
        acc = xyzzq_dbl_n(q, acc, c);
        xyzz o = wsum[w];
        xyzzq_add(q, acc, o);
    }
    if (q != 0) return;
    if (init) xyzz_madd(acc, affine_load(*init));
    if (out_raw) {
        out_raw[0] = acc;
        *reinterpret_cast<int*>(out_raw + 1) = *err_in;
        return;
    }
    affine_mem m = affine_store(xyzz_to_affine_var(acc));
    if (out_limbs) *out_limbs = m;
    if (out_be) {
        uint32_t* dst = reinterpret_cast<uint32_t*>(out_be);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            dst[7 - k] = __builtin_bswap32(m.x.w[k]);
            dst[15 - k] = __builtin_bswap32(m.y.w[k]);
        }
    }
}

// dst |= src (one lane): parks a call's input-validation flag for a later, synchronising call to report
__global__ void k_or_flag(int* __restrict__ dst, const int* __restrict__ src) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && *src) *dst = 1;
}
hipError_t launch_or_flag(int* dst, const int* src, hipStream_t st) {
    hipLaunchKernelGGL(k_or_flag, dim3(1), dim3(64), 0, st, dst, src);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// host-side launch helpers
size_t msm_slice_lds_bytes(int c, int M, int T) {
    int NB = 1 << (c - 1);
    return (size_t)T * sizeof(xyzz) + (size_t)2 * (NB + 1) * 4 + (size_t)M * 2;
}

hipError_t launch_ingest_points(const uint8_t* in_be, affine_mem* out, size_t n, int* err, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_ingest_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in_be, out, n, err);
    return hipGetLastError();
}
hipError_t launch_export_points(const affine_mem* in, uint8_t* out_be, size_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_export_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out_be, n);
    return hipGetLastError();
}
hipError_t launch_glv_expand(const affine_mem* in, affine_mem* lam, size_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_glv_expand, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, lam, n);
    return hipGetLastError();
}
hipError_t launch_glv_digits(const uint8_t* scalars_be, size_t n, size_t npad, int c, int W, int16_t* digits, hipStream_t st) {
    hipLaunchKernelGGL(k_glv_digits, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, st, scalars_be, n, npad, c, W, digits);
    return hipGetLastError();
}
hipError_t launch_scalars_to_digits(const uint8_t* scalars_be, size_t n, size_t npad, int c, int W, int16_t* digits, hipStream_t st) {
    hipLaunchKernelGGL(k_scalars_to_digits, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, st, scalars_be, n, npad, c, W, digits);
    return hipGetLastError();
}
hipError_t launch_msm_slice(const affine_mem* pts, const affine_mem* lam, size_t n1, const int16_t* digits, size_t n, size_t npad, int M, int c,
                            int W, int S, int T, xyzz* partial, hipStream_t st) {
    size_t lds = msm_slice_lds_bytes(c, M, T);
    dim3 grid((unsigned)S, (unsigned)W);
    if (T == 64)
        hipLaunchKernelGGL(k_msm_slice<64>, grid, dim3(64), lds, st, pts, lam, n1, digits, n, npad, M, c, partial);
    else if (T == 128)
        hipLaunchKernelGGL(k_msm_slice<128>, grid, dim3(128), lds, st, pts, lam, n1, digits, n, npad, M, c, partial);
    else
        hipLaunchKernelGGL(k_msm_slice<256>, grid, dim3(256), lds, st, pts, lam, n1, digits, n, npad, M, c, partial);
    return hipGetLastError();
}
hipError_t launch_msm_reduce_slices(const xyzz* partial, int S, int W, xyzz* wsum, hipStream_t st) {
    hipLaunchKernelGGL(k_msm_reduce_slices, dim3((unsigned)W), dim3(64), 0, st, partial, S, wsum);
    return hipGetLastError();
}
hipError_t launch_msm_reduce_parts(const xyzz* parts, int G, int W, xyzz* wsum, hipStream_t st) {
    hipLaunchKernelGGL(k_msm_reduce_parts, dim3((unsigned)W), dim3(64), 0, st, parts, G, W, wsum);
    return hipGetLastError();
}
hipError_t launch_msm_horner(const xyzz* wsum, int W, int c, const affine_mem* init, affine_mem* out_limbs, uint8_t* out_be, hipStream_t st,
                             xyzz* out_raw, const int* err_in) {
    hipLaunchKernelGGL(k_msm_horner, dim3(1), dim3(64), 0, st, wsum, W, c, init, out_limbs, out_be, out_raw, err_in);
    return hipGetLastError();
}

}  // namespace bp

BP_OPCOUNT_READER(bp_opcount_msm_kernels)
