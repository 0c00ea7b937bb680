// Random-linear-combination batch verification: N `aggregated_verify` checks
//     sum_p s_{j,p} G_p + sum_v t_{j,v} P_{j,v} == A_j        (/root/reference/src/proofs/range_proof.rs:742-777)
// hold for every j iff (except with probability ~2^-128 over the weights rho_j)
//     sum_p (sum_j rho_j s_{j,p}) G_p  +  sum_j sum_v (rho_j t_{j,v}) P_{j,v}  -  sum_j rho_j A_j  ==  O.
// The generator terms of all proofs collapse into ONE 2nm+2-term fixed-base MSM (scalars summed mod q,
// k_rlc_colsum), and the 17 proof-specific points per proof (A, S, T1, T2, L_j, R_j, V_j) of the whole batch
// go through one large Pippenger MSM (msm_sorted.hip) instead of one GLV ladder each.  When the combined
// check fails the caller falls back to the per-proof kernel, so verdicts are always the reference's.
//
//   k_rlc_weights      rho_j = low 128 bits of SHA-256("bpmsm-rlc-v1" || seed || be64(j)), 0 for malformed proofs
//   k_rlc_var_scalars  one lane per proof point (layout of k_bpv_decompress: A S T1 T2 L[k] R[k] V[m]):
//                      rho_j * t_{j,v}  (-rho_j for A_j) as 32-byte big-endian scalars for the MSM front end
//   k_rlc_colsum       (generator p, chunk of proofs) workgroups: sum_j rho_j s_{j,p} in Montgomery form,
//                      LDS tree; the last pass adds the chunk sums and writes big-endian scalars
//   k_rlc_verdicts     result == O ? (0 / 255 per proof) : all_ok = 0
#include <hip/hip_runtime.h>

#include "ec.h"
#include "fq.h"
#include "kernels.h"
#include "sha256.h"

namespace bp {

__global__ void k_rlc_weights(const uint8_t* __restrict__ seed32, size_t nproofs, const int* __restrict__ bad, fqm* __restrict__ rho) {
    size_t pr = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pr >= nproofs) return;
    if (bad[pr]) {
        rho[pr] = fqm_zero();
        return;
    }
    const char tag[12] = {'b', 'p', 'm', 's', 'm', '-', 'r', 'l', 'c', '-', 'v', '1'};
    uint8_t msg[64];
    for (int i = 0; i < 12; i++) msg[i] = (uint8_t)tag[i];
    for (int i = 0; i < 32; i++) msg[12 + i] = seed32[i];
    for (int i = 0; i < 8; i++) msg[44 + i] = (uint8_t)((uint64_t)pr >> (56 - 8 * i));
    // single padded block: 52 message bytes
    msg[52] = 0x80;
    for (int i = 53; i < 62; i++) msg[i] = 0;
    msg[62] = (uint8_t)((52 * 8) >> 8);
    msg[63] = (uint8_t)(52 * 8);
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    sha256_compress(h, msg);
    fq r = fq_zero();
    r.v[0] = h[7];
    r.v[1] = h[6];
    r.v[2] = h[5];
    r.v[3] = h[4];
    if ((r.v[0] | r.v[1] | r.v[2] | r.v[3]) == 0) r.v[0] = 1;
    rho[pr] = fqm_from_fq(r);
}

// Bulletproofs+ (wip): point order b_tag, A, L.., R.., a_tag, V.. (k_wipv_decompress), varS in the order of pts[1..]
__global__ void k_rlc_var_scalars(int k, int m, size_t nproofs, const fqm* __restrict__ rho, const fq* __restrict__ varS,
                                  uint8_t* __restrict__ sc_be, int wip) {
    const int NPT = (wip ? 3 : 4) + 2 * k + m, NV = (wip ? 2 : 3) + 2 * k + m;
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nproofs * (size_t)NPT) return;
    size_t pr = gid / NPT;
    int j = (int)(gid % NPT);
    fqm r = rho[pr];
    fq out;
    if (fqm_is_zero(r)) {
        out = fq_zero();
    } else if (j == 0) {
        out = fqm_to_fq(fqm_neg(r));  // - rho_j A_j  (Bulletproofs+: - rho_j b_tag_j)
    } else if (wip) {
        out = fqm_to_fq(fqm_mul(r, fqm_from_fq(varS[pr * (size_t)NV + (j - 1)])));
    } else {
        // point order A S T1 T2 L.. R.. V..  ->  varS order S, L.., R.., V.., T1, T2 (k_bpv_consts)
        int v = (j == 1) ? 0 : (j == 2) ? (1 + 2 * k + m) : (j == 3) ? (2 + 2 * k + m) : (j - 3);
        out = fqm_to_fq(fqm_mul(r, fqm_from_fq(varS[pr * (size_t)NV + v])));
    }
    fq_to_be(sc_be + 32 * gid, out);
}

#define RLC_T 256
#define RLC_CHUNK 4096  // proofs per workgroup
__global__ __launch_bounds__(RLC_T) void k_rlc_colsum(int np, size_t nproofs, const fqm* __restrict__ rho, const fqm* __restrict__ scF,
                                                      const int* __restrict__ bad, fqm* __restrict__ part) {
    __shared__ fqm red[RLC_T];
    const int tid = threadIdx.x, p = blockIdx.x, chunk = blockIdx.y;
    const size_t j0 = (size_t)chunk * RLC_CHUNK;
    const size_t j1 = j0 + RLC_CHUNK < nproofs ? j0 + RLC_CHUNK : nproofs;
    fqm acc = fqm_zero();
    for (size_t j = j0 + tid; j < j1; j += RLC_T) {
        if (bad[j]) continue;
        acc = fqm_add(acc, fqm_mul(rho[j], scF[j * (size_t)np + p]));
    }
    red[tid] = acc;
    __syncthreads();
    for (int off = RLC_T / 2; off >= 1; off >>= 1) {
        if (tid < off) red[tid] = fqm_add(red[tid], red[tid + off]);
        __syncthreads();
    }
    if (tid == 0) part[(size_t)p * gridDim.y + chunk] = red[0];
}
__global__ void k_rlc_colsum_final(int np, int nchunks, const fqm* __restrict__ part, uint8_t* __restrict__ out_be) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= np) return;
    fqm acc = fqm_zero();
    for (int ch = 0; ch < nchunks; ch++) acc = fqm_add(acc, part[(size_t)p * nchunks + ch]);
    fq_to_be(out_be + 32 * (size_t)p, fqm_to_fq(acc));
}

__global__ void k_rlc_verdicts(size_t nproofs, const uint8_t* __restrict__ result_be, const int* __restrict__ bad, uint8_t* __restrict__ verdicts,
                               int* __restrict__ all_ok) {
    size_t pr = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t o = 0;
    const uint32_t* r = reinterpret_cast<const uint32_t*>(result_be);
    for (int i = 0; i < 16; i++) o |= r[i];
    if (pr == 0) *all_ok = (o == 0) ? 1 : 0;
    if (pr >= nproofs || o != 0) return;
    verdicts[pr] = bad[pr] ? 255 : 0;
}

hipError_t launch_rlc_weights(const uint8_t* seed32, size_t nproofs, const int* bad, fqm* rho, hipStream_t st) {
    hipLaunchKernelGGL(k_rlc_weights, dim3((unsigned)((nproofs + 255) / 256)), dim3(256), 0, st, seed32, nproofs, bad, rho);
    return hipGetLastError();
}
hipError_t launch_rlc_var_scalars(int k, int m, size_t nproofs, const fqm* rho, const fq* varS, uint8_t* sc_be, hipStream_t st, int wip) {
    size_t total = nproofs * (size_t)((wip ? 3 : 4) + 2 * k + m);
    hipLaunchKernelGGL(k_rlc_var_scalars, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, k, m, nproofs, rho, varS, sc_be, wip);
    return hipGetLastError();
}
size_t rlc_colsum_chunks(size_t nproofs) { return (nproofs + RLC_CHUNK - 1) / RLC_CHUNK; }
hipError_t launch_rlc_colsum(int np, size_t nproofs, const fqm* rho, const fqm* scF, const int* bad, fqm* part, uint8_t* out_be, hipStream_t st) {
    const unsigned nchunks = (unsigned)rlc_colsum_chunks(nproofs);
    hipLaunchKernelGGL(k_rlc_colsum, dim3((unsigned)np, nchunks), dim3(RLC_T), 0, st, np, nproofs, rho, scF, bad, part);
    hipLaunchKernelGGL(k_rlc_colsum_final, dim3((unsigned)((np + 63) / 64)), dim3(64), 0, st, np, (int)nchunks, part, out_be);
    return hipGetLastError();
}
hipError_t launch_rlc_verdicts(size_t nproofs, const uint8_t* result_be, const int* bad, uint8_t* verdicts, int* all_ok, hipStream_t st) {
    hipLaunchKernelGGL(k_rlc_verdicts, dim3((unsigned)((nproofs + 255) / 256)), dim3(256), 0, st, nproofs, result_be, bad, verdicts, all_ok);
    return hipGetLastError();
}

}  // namespace bp
