// Fixed-base precomputation for the shared generators (g_vec, h_vec, G, H).
//
// In the reference every call receives `g_vec / h_vec / G / H` slices and multiplies each of them by
// a fresh 256-bit scalar (e.g. /root/reference/src/proofs/range_proof.rs:742-774: 2nm+1 of the
// 2nm+2k+m+5 terms of aggregated_verify are these fixed points).  MI355X has 288 GB of HBM and a
// 256 MiB Infinity Cache, so the generators are expanded ONCE into a table
//     T[p][w][d-1] = d * 2^(c*w) * P_p      (affine, 64 B),  d = 1 .. 2^(c-1),  w = 0 .. 256/c
// and a fixed-base term becomes W gathers + W mixed additions with no doublings at all.
// Build: per-point doubling chain -> per-(p,w) running multiples (XYZZ, temp) -> chunked
// Montgomery batch inversion to affine.
#include <hip/hip_runtime.h>

#include "ec.h"
#include "kernels.h"

namespace bp {

// bases[p*W + w] = 2^(c*w) * P_p  (XYZZ); one thread per point (serial doubling chain).
__global__ void k_gens_bases(const affine_mem* __restrict__ pts, int np, int c, int W, xyzz* __restrict__ bases) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= np) return;
    xyzz acc = xyzz_from_affine(affine_load(pts[p]));
    bases[(size_t)p * W] = acc;
    for (int w = 1; w < W; w++) {
        for (int k = 0; k < c; k++) acc = xyzz_dbl(acc);
        bases[(size_t)p * W + w] = acc;
    }
}

// out[i] = affine(in[i]); one thread per element, one inversion each (small arrays only).
__global__ void k_xyzz_to_affine(const xyzz* __restrict__ in, affine_mem* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = affine_store(xyzz_to_affine(in[i]));
}

// tmp[idx*NB + d-1] = d * B_idx, d = 1..NB; one thread per (idx = p*W + w, segment of NB/segs multiples):
// a segment starts from (seg * NB/segs) * B by double-and-add, so wide windows (NB = 2^19) do not leave the
// GPU on a handful of half-million-step serial chains.
__global__ void k_gens_multiples(const affine_mem* __restrict__ basesA, size_t nidx, int NB, int segs, xyzz* __restrict__ tmp) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t idx = g / segs;
    const int seg = (int)(g % segs);
    if (idx >= nidx) return;
    affine B = affine_load(basesA[idx]);
    const int S = NB / segs;
    const uint32_t k0 = (uint32_t)seg * (uint32_t)S;
    xyzz acc = xyzz_inf();
    for (int b = 31; b >= 0; b--) {
        acc = xyzz_dbl(acc);
        if ((k0 >> b) & 1) xyzz_madd(acc, B);
    }
    xyzz* dst = tmp + idx * (size_t)NB + k0;
    for (int d = 0; d < S; d++) {
        xyzz_madd(acc, B);
        dst[d] = acc;
    }
}

// Chunked batch normalisation: each thread converts CH consecutive XYZZ entries to affine with one
// field inversion (Montgomery's trick on the ZZZ coordinates).  Entries must not be infinity.
#define GENS_CH 16
__global__ void k_gens_normalize(const xyzz* __restrict__ tmp, size_t n, affine_mem* __restrict__ table) {
    size_t chunk = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t i0 = chunk * GENS_CH;
    if (i0 >= n) return;
    int cnt = (int)((n - i0) < GENS_CH ? (n - i0) : GENS_CH);
    fp pref[GENS_CH];
    fp run = fp_one();
    for (int i = 0; i < cnt; i++) {
        run = fp_mul(run, tmp[i0 + i].ZZZ);
        pref[i] = run;
    }
    fp inv = fp_inv(run);
    for (int i = cnt - 1; i >= 0; i--) {
        xyzz e = tmp[i0 + i];
        fp zi3 = (i == 0) ? inv : fp_mul(inv, pref[i - 1]);
        inv = fp_mul(inv, e.ZZZ);
        fp zi = fp_mul(e.ZZ, zi3);
        fp zi2 = fp_sqr(zi);
        affine a;
        a.x = fp_mul(e.X, zi2);
        a.y = fp_mul(e.Y, zi3);
        table[i0 + i] = affine_store(a);
    }
}

hipError_t launch_gens_bases(const affine_mem* pts, int np, int c, int W, xyzz* bases, hipStream_t st) {
    hipLaunchKernelGGL(k_gens_bases, dim3((np + 63) / 64), dim3(64), 0, st, pts, np, c, W, bases);
    return hipGetLastError();
}
hipError_t launch_xyzz_to_affine(const xyzz* in, affine_mem* out, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_xyzz_to_affine, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, in, out, n);
    return hipGetLastError();
}
hipError_t launch_gens_multiples(const affine_mem* basesA, size_t nidx, int NB, xyzz* tmp, hipStream_t st) {
    int segs = NB / 2048;  // chains of at most 2048 additions (the rows come in groups, verify_api.cpp: enough chains per group to fill the SIMDs)
    if (segs < 1) segs = 1;
    const size_t lanes = nidx * (size_t)segs;
    hipLaunchKernelGGL(k_gens_multiples, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, st, basesA, nidx, NB, segs, tmp);
    return hipGetLastError();
}
hipError_t launch_gens_normalize(const xyzz* tmp, size_t n, affine_mem* table, hipStream_t st) {
    size_t chunks = (n + GENS_CH - 1) / GENS_CH;
    hipLaunchKernelGGL(k_gens_normalize, dim3((unsigned)((chunks + 63) / 64)), dim3(64), 0, st, tmp, n, table);
    return hipGetLastError();
}

}  // namespace bp
