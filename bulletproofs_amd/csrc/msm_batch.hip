// bp_msm_batch: many small MSMs that share the generator prefix - the shape of the
// reference's fast verifiers when many statements are checked at once:
//     out_j = sum_p gen_scalars[j][p] * gens[p]  +  sum_e extra_scalars[j][e] * extra_points[j][e]
// (`InnerProductArg::fast_verify`, /root/reference/src/proofs/inner_product.rs:334-338;
//  `RangeProof::aggregated_verify`, range_proof.rs:770-774; weighted_inner_product.rs:500-504).
//
//   k_mb_digits : one lane per (job, generator): scalar -> signed window digits of the table's width
//   k_mb_msm    : 16 lanes per job: generator terms through table gathers (no doublings), job-specific
//                 points through GLV + 4-bit windows, lane tree sum -> XYZZ per job
// followed by the shared batched normalisation (k_xyzz_to_affine) and export.
#include <hip/hip_runtime.h>

#include "ec.h"
#include "ecmul.h"
#include "ecwave.h"
#include "fq.h"
#include "kernels.h"

namespace bp {

// mont != 0: the 32 bytes are the host scalar class's Montgomery limbs (hostmath.h Fq: little-endian words of x R mod q,
// R = 2^256) and the conversion to the plain value happens here, one lane per scalar, instead of one Montgomery product
// per term on a host thread (the protocol layer uploads thousands of terms per call).
__global__ void k_mb_digits(const uint8_t* __restrict__ scalars_be, size_t total, int c, int W, tdig_t* __restrict__ digits, int mont) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    fq k;
    if (mont) {
        const uint32_t* wsrc = reinterpret_cast<const uint32_t*>(scalars_be + 32 * i);
#pragma unroll
        for (int j = 0; j < 8; j++) k.v[j] = wsrc[j];
        k = fq_from_mont(k);
    } else {
        k = fq_from_be_reduce(scalars_be + 32 * i);
    }
    tdig_t* out = digits + i * (size_t)W;
    uint32_t carry = 0;
    const uint32_t half = 1u << (c - 1), mask = (1u << c) - 1;
    for (int w = 0; w < W; w++) {
        int bit = w * c;
        uint32_t v = 0;
        if (bit < 256) {
            v = fq_bits32(k.v, bit) & mask;
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
        out[w] = (tdig_t)d;
    }
}

#define MB_TP 16
__global__ __launch_bounds__(256, 2) void k_mb_msm(size_t njobs, int np, const affine_mem* __restrict__ table, int c, int W,
                                                   const tdig_t* __restrict__ digF, const uint8_t* __restrict__ has_gen,
                                                   const affine_mem* __restrict__ xpts, const uint8_t* __restrict__ xsc_be,
                                                   const uint32_t* __restrict__ xoff, xyzz* __restrict__ vartab, affine_mem* __restrict__ varaff, int8_t* __restrict__ vardig,
                                                   xyzz* __restrict__ out) {
    const int tid = threadIdx.x, lane = tid % MB_TP;
    const size_t job = (size_t)blockIdx.x * (256 / MB_TP) + tid / MB_TP;
    const int NB = 1 << (c - 1);
    xyzz acc = xyzz_inf();
    if (job < njobs) {
        if (has_gen[job]) {
            const int nfw = np * W;
            const tdig_t* dg = digF + job * (size_t)nfw;
            for (int idx = lane; idx < nfw; idx += MB_TP) {
                int d = dg[idx];
                if (d != 0) {
                    affine e = affine_load(table[(size_t)idx * NB + ((d < 0 ? -d : d) - 1)]);
                    xyzz_madd_s(acc, e, d < 0);
                }
            }
        }
        const size_t gl = (size_t)blockIdx.x * 256 + tid;
        xyzz* tab = vartab + gl * ECMUL_SLOTS;
        affine_mem* atab = varaff + gl * ECMUL_NT;
        int8_t* dig = vardig + gl * ECMUL_DIGB;
        for (uint32_t v = xoff[job] + lane; v < xoff[job + 1]; v += MB_TP) {
            xyzz t = ec_mul_glv_multi<1>([&](int) { return affine_load(xpts[v]); }, [&](int) { return fq_from_be_reduce(xsc_be + 32 * (size_t)v); }, 1,
                                         tab, atab, dig);
            xyzz_add(acc, t);
        }
    }
    xyzz_wave_tree<MB_TP>(acc, lane);  // DPP row shifts (ecwave.h)
    if (lane == 0 && job < njobs) out[job] = acc;
}

hipError_t launch_mb_digits(const uint8_t* scalars_be, size_t total, int c, int W, tdig_t* digits, hipStream_t st, int mont) {
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_mb_digits, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, scalars_be, total, c, W, digits, mont);
    return hipGetLastError();
}
size_t mb_vartab_bytes(size_t njobs) {
    size_t blocks = (njobs + (256 / MB_TP) - 1) / (256 / MB_TP);
    return blocks * 256 * (ECMUL_SLOTS * sizeof(xyzz) + ECMUL_NT * sizeof(affine_mem) + ECMUL_DIGB);
}
hipError_t launch_mb_msm(size_t njobs, int np, const affine_mem* table, int c, int W, const tdig_t* digF, const uint8_t* has_gen,
                         const affine_mem* xpts, const uint8_t* xsc_be, const uint32_t* xoff, xyzz* vartab, xyzz* out, hipStream_t st) {
    unsigned blocks = (unsigned)((njobs + (256 / MB_TP) - 1) / (256 / MB_TP));
    affine_mem* varaff = reinterpret_cast<affine_mem*>(vartab + (size_t)blocks * 256 * ECMUL_SLOTS);
    int8_t* vardig = reinterpret_cast<int8_t*>(varaff + (size_t)blocks * 256 * ECMUL_NT);
    hipLaunchKernelGGL(k_mb_msm, dim3(blocks), dim3(256), 0, st, njobs, np, table, c, W, digF, has_gen, xpts, xsc_be, xoff, vartab, varaff, vardig,
                       out);
    return hipGetLastError();
}

}  // namespace bp
