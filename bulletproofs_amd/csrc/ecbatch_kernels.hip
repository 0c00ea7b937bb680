// Batched independent point operations for the prover / slow verifiers (one lane per element):
//   k_scalar_mul_batch : out[i] = k_i * P_i          hi_tag[i] = y^-i * h_i
//                        (/root/reference/src/proofs/range_proof.rs:231-233, 327-329, 462-464)
//   k_fold_bases       : out[i] = s_lo * lo[i] + s_hi * hi[i]   with SHARED scalars
//                        G_new / H_new of the inner-product rounds (inner_product.rs:140-155, 203-218;
//                        weighted_inner_product.rs:189-204, 312-327): every lane follows the same
//                        digit schedule, so the wavefront never diverges.
// Both use the GLV fixed-window ladder of ecmul.h and normalise with one inversion per lane.
#include <hip/hip_runtime.h>

#include "ec.h"
#include "ecmul.h"
#include "fq.h"
#include "kernels.h"

namespace bp {

__device__ __forceinline__ fq load_scalar_be(const uint8_t* __restrict__ p) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(p);
    fq k;
#pragma unroll
    for (int j = 0; j < 8; j++) k.v[j] = __builtin_bswap32(src[7 - j]);
    fq_final_reduce(k.v, 0);
    return k;
}

__global__ __launch_bounds__(256, 2) void k_scalar_mul_batch(const affine_mem* __restrict__ pts, const uint8_t* __restrict__ scalars_be, size_t n,
                                                             affine_mem* __restrict__ out, xyzz* __restrict__ vartab) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fq k = load_scalar_be(scalars_be + 32 * i);
    xyzz r = ec_mul_glv(affine_load(pts[i]), k, vartab + i * ECMUL_NT);
    out[i] = affine_store(xyzz_to_affine(r));
}

__global__ __launch_bounds__(256, 2) void k_fold_bases(const affine_mem* __restrict__ lo, const affine_mem* __restrict__ hi, size_t n_half,
                                                       const uint8_t* __restrict__ s_lo_be, const uint8_t* __restrict__ s_hi_be,
                                                       affine_mem* __restrict__ out, xyzz* __restrict__ vartab) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_half) return;
    fq slo = load_scalar_be(s_lo_be), shi = load_scalar_be(s_hi_be);
    xyzz a = ec_mul_glv(affine_load(lo[i]), slo, vartab + i * ECMUL_NT);
    xyzz b = ec_mul_glv(affine_load(hi[i]), shi, vartab + i * ECMUL_NT);
    xyzz_add(a, b);
    out[i] = affine_store(xyzz_to_affine(a));
}

size_t ecbatch_vartab_bytes(size_t n) { return n * ECMUL_NT * sizeof(xyzz); }

hipError_t launch_scalar_mul_batch(const affine_mem* pts, const uint8_t* scalars_be, size_t n, affine_mem* out, xyzz* vartab, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scalar_mul_batch, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, pts, scalars_be, n, out, vartab);
    return hipGetLastError();
}
hipError_t launch_fold_bases(const affine_mem* lo, const affine_mem* hi, size_t n_half, const uint8_t* s_lo_be, const uint8_t* s_hi_be,
                             affine_mem* out, xyzz* vartab, hipStream_t st) {
    if (n_half == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fold_bases, dim3((unsigned)((n_half + 63) / 64)), dim3(64), 0, st, lo, hi, n_half, s_lo_be, s_hi_be, out, vartab);
    return hipGetLastError();
}

}  // namespace bp
