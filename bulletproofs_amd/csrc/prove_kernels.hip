// Scalar side of the BATCH prover `RangeProof::prove` x N on the device
// (/root/reference/src/proofs/range_proof.rs:45-263, inner_product.rs:36-169): every per-proof vector of the reference's
// single-threaded prover - the bit vectors aL / aR (:81-83), yi (:131-133), the polynomial coefficients t1 / t2 (:135-162),
// Lp / Rp / tx (:186-211), and, per inner-product round, c_L / c_R (:64-65), the folds of a and b (:122-137) and the
// coefficient vectors that replace the folded bases (G^(j)[i] = sum_{t = i mod n'} cG[t] g_t) - is a kernel
// over [proof][i] on the reduced-radix Montgomery form `fqm` (fq.h), the way verify_scalars.hip does it for the verifier.
// The kernels write the terms of the table MSMs directly, as compact lists of (table slot, signed window digit) entries
// - only the rows a step uses: 77 entries for A, (nm + 1) W for an L or R, W for a one-term job - which k_pv_msm sums (two
// jobs per proof); the host keeps only the Fiat-Shamir transcript (SHA-256 of the points each step returns) and the
// proof encoding.  Same values mod q, same group elements, hence the same proof bytes as the host path
// (tests/test_gpu_proto.py::test_prove_batch_*).
#include <hip/hip_runtime.h>

#include "ec.h"
#include "ecwave.h"
#include "fq.h"
#include "kernels.h"

namespace bp {

#define PV_T 256  // most lanes of a per-proof workgroup (strided over i; 64 or 128 for short vectors: pv_lanes)
static unsigned pv_lanes(int nm) { return nm <= 64 ? 64u : (nm <= 128 ? 128u : (unsigned)PV_T); }

// the W signed window digits of scalar kk on table row `row`, as entries (slot = row W + w, digit)
__device__ __forceinline__ void pv_store_digits(const fq& kk, int c, int W, int row, pv_ent* __restrict__ out) {
    const uint32_t half = 1u << (c - 1), mask = (1u << c) - 1;
    uint32_t carry = 0;
    for (int w = 0; w < W; w++) {
        const int bit = w * c;
        uint32_t v = 0;
        if (bit < 256) {
            v = fq_bits32(kk.v, bit) & mask;
        }
        v += carry;
        pv_ent e;
        e.slot = (uint32_t)(row * W + w);
        if (v > half) {
            carry = 1;
            e.d = (int)v - (int)(1u << c);
        } else {
            carry = 0;
            e.d = (int)v;
        }
        out[w] = e;
    }
}
__device__ __forceinline__ void pv_digits(const fqm& s, int c, int W, int row, pv_ent* __restrict__ out) {
    pv_store_digits(fqm_to_fq(s), c, W, row, out);
}
__device__ __forceinline__ fqm pv_be(const uint8_t* be32) { return fqm_from_fq(fq_from_be_reduce(be32)); }
__device__ __forceinline__ fqm pv_pow(fqm base, uint32_t e) {
    fqm r = fqm_one();
    while (e) {
        if (e & 1u) r = fqm_mul(r, base);
        e >>= 1;
        if (e) base = fqm_mul(base, base);
    }
    return r;
}
// 2^r (r < 261) in Montgomery form: the plain limb pattern times R^2
__device__ __forceinline__ fqm pv_two_pow(int r) {
    fqm t = fqm_zero();
    t.v[r / 29] = 1u << (r % 29);
    return fqm_mul(t, fqm_r2());
}
// sum of v over the workgroup's lanes, returned to every lane
__device__ __forceinline__ fqm pv_block_sum(fqm v, fqm* red) {
    const int tid = threadIdx.x;
    __syncthreads();
    red[tid] = v;
    __syncthreads();
    for (int off = (int)blockDim.x / 2; off >= 1; off >>= 1) {
        if (tid < off) red[tid] = fqm_add(red[tid], red[tid + off]);
        __syncthreads();
    }
    return red[0];
}

// Tape of a proof (32-byte big-endian scalars in the reference's draw order, range_proof.rs:57-58, 109-114, 164-165):
//   alpha, rho, SR[0..nm), SL[0..nm), tau1, tau2
#define TP_ALPHA 0
#define TP_RHO 1
#define TP_SR(i) (2 + (i))
#define TP_SL(nm, i) (2 + (nm) + (i))
#define TP_TAU1(nm) (2 + 2 * (nm))
#define TP_TAU2(nm) (3 + 2 * (nm))

// A = alpha H + sum_{bit} g_i - sum_{!bit} h_i (:93-107) and S = rho H + sum SL_i g_i + SR_i h_i (:116-122): entries of the
// two jobs of proof pr (jobs 2 pr, 2 pr + 1; nm + W and (2 nm + 1) W entries); SL, SR kept as fqm for the later steps.
__global__ __launch_bounds__(PV_T) void k_pv_init(int nm, int c, int W, const uint8_t* __restrict__ tapes, size_t tape_len,
                                                  const uint8_t* __restrict__ bits, fqm* __restrict__ SL, fqm* __restrict__ SR,
                                                  pv_ent* __restrict__ ent, size_t stride) {
    const size_t pr = blockIdx.x;
    const uint8_t* tp = tapes + pr * tape_len;
    pv_ent* eA = ent + (2 * pr) * stride;
    pv_ent* eS = ent + (2 * pr + 1) * stride;
    for (int i = threadIdx.x; i < nm; i += (int)blockDim.x) {
        const fqm sr = pv_be(tp + 32 * (size_t)TP_SR(i)), sl = pv_be(tp + 32 * (size_t)TP_SL(nm, i));
        SR[pr * nm + i] = sr;
        SL[pr * nm + i] = sl;
        const int bit = bits[pr * nm + i];
        pv_ent u;  // + g_i or - h_i: digit +-1 of window 0
        u.slot = (uint32_t)((bit ? i : nm + i) * W);
        u.d = bit ? 1 : -1;
        eA[i] = u;
        pv_digits(sl, c, W, i, eS + (size_t)i * W);
        pv_digits(sr, c, W, nm + i, eS + (size_t)(nm + i) * W);
    }
    if (threadIdx.x == 0) {
        pv_digits(pv_be(tp + 32 * TP_ALPHA), c, W, 2 * nm + 1, eA + nm);
        pv_digits(pv_be(tp + 32 * TP_RHO), c, W, 2 * nm + 1, eS + (size_t)(2 * nm) * W);
    }
}

// one job per proof with ONE term: scalars[pr] times table row `row` (y * generator :126, x_u * G :219)
__global__ void k_pv_term(size_t nproofs, int c, int W, int row, const uint8_t* __restrict__ scalars_be, pv_ent* __restrict__ ent, size_t stride) {
    const size_t pr = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pr >= nproofs) return;
    pv_digits(pv_be(scalars_be + 32 * pr), c, W, row, ent + pr * stride);
}

// yi, z powers, t1, t2 (:131-162) and the digits of T1 = t1 G + tau1 H, T2 = t2 G + tau2 H (:168-169); y^-1 for hi_tag
__global__ __launch_bounds__(PV_T) void k_pv_poly(int n_bits, int m, int c, int W, const uint8_t* __restrict__ tapes, size_t tape_len,
                                                  const uint8_t* __restrict__ yz_be /* [N][2][32] */, const uint8_t* __restrict__ bits,
                                                  const fqm* __restrict__ SL, const fqm* __restrict__ SR, fqm* __restrict__ yi,
                                                  fqm* __restrict__ zpow /* [N][m] */, fqm* __restrict__ yinv, pv_ent* __restrict__ ent, size_t stride) {
    __shared__ fqm red[PV_T];
    const size_t pr = blockIdx.x;
    const int nm = n_bits * m;
    const fqm y = pv_be(yz_be + 64 * pr), z = pv_be(yz_be + 64 * pr + 32), ONE = fqm_one();
    fqm t1 = fqm_zero(), t2 = fqm_zero();
    for (int i = threadIdx.x; i < nm; i += (int)blockDim.x) {
        const fqm yp = pv_pow(y, (uint32_t)i);
        yi[pr * nm + i] = yp;
        const fqm sl = SL[pr * nm + i], sr = SR[pr * nm + i];
        const fqm aL = bits[pr * nm + i] ? ONE : fqm_zero(), aR = fqm_sub(aL, ONE);
        const fqm sry = fqm_mul(sr, yp);
        t2 = fqm_add(t2, fqm_mul(sry, sl));                                              // :135-137
        const fqm zj2k = fqm_mul(pv_pow(z, 2u + (uint32_t)(i / n_bits)), pv_two_pow(i % n_bits));
        fqm term = fqm_mul(sl, fqm_mul(fqm_add(aR, z), yp));
        term = fqm_add(term, fqm_mul(fqm_sub(aL, z), sry));
        term = fqm_add(term, fqm_mul(zj2k, sl));
        t1 = fqm_add(t1, term);                                                          // :145-162
    }
    t1 = pv_block_sum(t1, red);
    t2 = pv_block_sum(t2, red);
    pv_ent* e1 = ent + (2 * pr) * stride;  // 2 W entries each
    pv_ent* e2 = ent + (2 * pr + 1) * stride;
    if (threadIdx.x == 0) {
        const uint8_t* tp = tapes + pr * tape_len;
        pv_digits(t1, c, W, 2 * nm, e1);
        pv_digits(pv_be(tp + 32 * (size_t)TP_TAU1(nm)), c, W, 2 * nm + 1, e1 + W);
        pv_digits(t2, c, W, 2 * nm, e2);
        pv_digits(pv_be(tp + 32 * (size_t)TP_TAU2(nm)), c, W, 2 * nm + 1, e2 + W);
        fqm zp = fqm_mul(z, z);
        for (int j = 0; j < m; j++) {
            zpow[pr * m + j] = zp;
            zp = fqm_mul(zp, z);
        }
        yinv[pr] = fqm_inv_var(y);  // (y = 0 is refused by the host before this launch)
    }
}

// Lp, Rp, tx, tau_x, miu (:171-212) -> a, b of the inner-product argument; cG = 1, cH = y^-i (hi_tag, :223-233)
__global__ __launch_bounds__(PV_T) void k_pv_lr(int n_bits, int m, const uint8_t* __restrict__ tapes, size_t tape_len,
                                                const uint8_t* __restrict__ yz_be, const uint8_t* __restrict__ x_be,
                                                const uint8_t* __restrict__ blindings_be, const uint8_t* __restrict__ bits,
                                                const fqm* __restrict__ SL, const fqm* __restrict__ SR, const fqm* __restrict__ yi,
                                                const fqm* __restrict__ zpow, const fqm* __restrict__ yinv, fqm* __restrict__ a, fqm* __restrict__ b,
                                                fqm* __restrict__ cG, fqm* __restrict__ cH, uint8_t* __restrict__ out_be /* [N][96] */) {
    __shared__ fqm red[PV_T];
    const size_t pr = blockIdx.x;
    const int nm = n_bits * m;
    const fqm z = pv_be(yz_be + 64 * pr + 32), x = pv_be(x_be + 32 * pr), ONE = fqm_one(), yv = yinv[pr];
    fqm tx = fqm_zero();
    for (int i = threadIdx.x; i < nm; i += (int)blockDim.x) {
        const fqm sl = SL[pr * nm + i], sr = SR[pr * nm + i], yp = yi[pr * nm + i];
        const fqm aL = bits[pr * nm + i] ? ONE : fqm_zero(), aR = fqm_sub(aL, ONE);
        const fqm Lp = fqm_add(fqm_mul(sl, x), fqm_sub(aL, z));                                           // :186-192
        const fqm zj2k = fqm_mul(zpow[pr * m + i / n_bits], pv_two_pow(i % n_bits));
        const fqm Rp = fqm_add(fqm_mul(yp, fqm_add(fqm_add(z, aR), fqm_mul(sr, x))), zj2k);               // :194-207
        a[pr * nm + i] = Lp;
        b[pr * nm + i] = Rp;
        tx = fqm_add(tx, fqm_mul(Lp, Rp));                                                                // :208-211
        cG[pr * nm + i] = ONE;
        cH[pr * nm + i] = pv_pow(yv, (uint32_t)i);
    }
    tx = pv_block_sum(tx, red);
    if (threadIdx.x == 0) {
        const uint8_t* tp = tapes + pr * tape_len;
        const fqm tau1 = pv_be(tp + 32 * (size_t)TP_TAU1(nm)), tau2 = pv_be(tp + 32 * (size_t)TP_TAU2(nm));
        fqm tau_x = fqm_add(fqm_mul(x, tau1), fqm_mul(fqm_mul(x, x), tau2));
        for (int j = 0; j < m; j++) tau_x = fqm_add(tau_x, fqm_mul(zpow[pr * m + j], pv_be(blindings_be + 32 * (pr * m + j))));   // :175-183
        const fqm miu = fqm_add(fqm_mul(pv_be(tp + 32 * TP_RHO), x), pv_be(tp + 32 * TP_ALPHA));                                // :184
        fq_to_be(out_be + 96 * pr, fqm_to_fq(tau_x));
        fq_to_be(out_be + 96 * pr + 32, fqm_to_fq(miu));
        fq_to_be(out_be + 96 * pr + 64, fqm_to_fq(tx));
    }
}

// One inner-product round over the ORIGINAL generators (n = current length, h = n / 2): c_L, c_R (inner_product.rs:64-65) and
// the entries of  L = <a_L, G_R> + <b_R, H_L> + c_L ux (:70-90),  R = <a_R, G_L> + <b_L, H_R> + c_R ux (:95-115)  with
// G^(j), H^(j) written through the coefficient vectors cG, cH; ux = xu * G, so its term is the scalar c xu on G's row.
__global__ __launch_bounds__(PV_T) void k_pv_round(int nm, int n, int c, int W, const uint8_t* __restrict__ xu_be, const fqm* __restrict__ a,
                                                   const fqm* __restrict__ b, const fqm* __restrict__ cG, const fqm* __restrict__ cH,
                                                   pv_ent* __restrict__ ent, size_t stride) {
    __shared__ fqm red[PV_T];
    const size_t pr = blockIdx.x;
    const int h = n / 2;
    const fqm* A = a + pr * nm;
    const fqm* B = b + pr * nm;
    fqm cL = fqm_zero(), cR = fqm_zero();
    for (int i = threadIdx.x; i < h; i += (int)blockDim.x) {
        cL = fqm_add(cL, fqm_mul(A[i], B[h + i]));
        cR = fqm_add(cR, fqm_mul(A[h + i], B[i]));
    }
    cL = pv_block_sum(cL, red);
    cR = pv_block_sum(cR, red);
    pv_ent* eL = ent + (2 * pr) * stride;  // (nm + 1) W entries each: term t of a job is its t-th generator row
    pv_ent* eR = ent + (2 * pr + 1) * stride;
    for (int t = threadIdx.x; t < nm; t += (int)blockDim.x) {
        const int r = t & (n - 1);
        const fqm g = cG[pr * nm + t], hh = cH[pr * nm + t];
        if (r >= h) {
            pv_digits(fqm_mul(A[r - h], g), c, W, t, eL + (size_t)t * W);            // <a_L, G_R>
            pv_digits(fqm_mul(B[r - h], hh), c, W, nm + t, eR + (size_t)t * W);      // <b_L, H_R>
        } else {
            pv_digits(fqm_mul(B[h + r], hh), c, W, nm + t, eL + (size_t)t * W);      // <b_R, H_L>
            pv_digits(fqm_mul(A[h + r], g), c, W, t, eR + (size_t)t * W);            // <a_R, G_L>
        }
    }
    if (threadIdx.x == 0) {
        const fqm xu = pv_be(xu_be + 32 * pr);
        pv_digits(fqm_mul(cL, xu), c, W, 2 * nm, eL + (size_t)nm * W);
        pv_digits(fqm_mul(cR, xu), c, W, 2 * nm, eR + (size_t)nm * W);
    }
}

// a' = x a_L + x^-1 a_R, b' = x b_R + x^-1 b_L (:122-137); the base folds G' = x^-1 G_L + x G_R, H' = x H_L + x^-1 H_R
// (:140-155) become updates of the coefficient vectors
__global__ __launch_bounds__(PV_T) void k_pv_fold(int nm, int n, const uint8_t* __restrict__ xj_be, fqm* __restrict__ a, fqm* __restrict__ b,
                                                  fqm* __restrict__ cG, fqm* __restrict__ cH) {
    __shared__ fqm xs[2];
    const size_t pr = blockIdx.x;
    const int h = n / 2;
    if (threadIdx.x == 0) {
        xs[0] = pv_be(xj_be + 32 * pr);
        xs[1] = fqm_inv_var(xs[0]);  // (x = 0 is refused by the host before this launch)
    }
    __syncthreads();
    const fqm x = xs[0], xi = xs[1];
    fqm* A = a + pr * nm;
    fqm* B = b + pr * nm;
    for (int i = threadIdx.x; i < h; i += (int)blockDim.x) {  // lane i reads and writes only its own pair of slots
        const fqm an = fqm_add(fqm_mul(A[i], x), fqm_mul(A[h + i], xi));
        const fqm bn = fqm_add(fqm_mul(B[h + i], x), fqm_mul(B[i], xi));
        A[i] = an;
        B[i] = bn;
    }
    if (h != 1)
        for (int t = threadIdx.x; t < nm; t += (int)blockDim.x) {
            const bool lower = (t & (n - 1)) < h;
            cG[pr * nm + t] = fqm_mul(cG[pr * nm + t], lower ? xi : x);
            cH[pr * nm + t] = fqm_mul(cH[pr * nm + t], lower ? x : xi);
        }
}

// The table MSMs of one step: job j sums its cnt (even jobs: cnt0, odd jobs: cnt1) entries ent[j stride ..], one table
// gather and one mixed addition per entry (the table of bp_gens: [slot][2^(c-1)] affine multiples), TP lanes per job with
// the next gather in flight behind the current addition (as bpv_fix_lane_sum, verify_kernels.hip), LDS tree, one XYZZ sum
// per (job, part): with few long jobs (aggregated proofs: 10^5 entries each) a job is dealt over `parts` workgroups of 256
// lanes, whose partial sums k_pv_msm_combine adds.
template <int TP>
__global__ __launch_bounds__(256, 3) void k_pv_msm(size_t njobs, const affine_mem* __restrict__ table, int c, const pv_ent* __restrict__ ent,
                                                   size_t stride, int cnt0, int cnt1, int parts, xyzz* __restrict__ out) {
    __shared__ xyzz red[TP > 64 ? 256 : 1];
    const int tid = threadIdx.x, lane = tid % TP;
    const size_t unit = (size_t)blockIdx.x * (256 / TP) + tid / TP;  // (job, part)
    const size_t job = unit / (size_t)parts;
    const int part = (int)(unit % (size_t)parts), step = TP * parts;
    xyzz acc = xyzz_inf();
    if (job < njobs) {
        const int n = (job & 1) ? cnt1 : cnt0;
        const pv_ent* e = ent + job * stride;
        const size_t NB = (size_t)1 << (c - 1);
        auto fetch = [&](int idx) { return e[idx < n ? idx : n - 1]; };
        auto gather = [&](const pv_ent& t) {
            const int a = t.d < 0 ? -t.d : t.d;
            return affine_mem_ld(table + (size_t)t.slot * NB + (a ? a - 1 : 0));
        };
        auto add = [&](const affine_mem& m, const pv_ent& t, int idx) {
            if (t.d != 0 && idx < n) {
                affine p = affine_load(m);
                xyzz_madd_s(acc, p, t.d < 0);
            }
        };
        const int first = lane + TP * part;
        pv_ent tA = fetch(first), tB = fetch(first + step);
        affine_mem A = gather(tA), B;
        for (int idx = first; idx < n; idx += 2 * step) {
            pv_ent tN = fetch(idx + 2 * step);
            B = gather(tB);
            add(A, tA, idx);
            tA = tN;
            tN = fetch(idx + 3 * step);
            A = gather(tA);
            add(B, tB, idx + step);
            tB = tN;
        }
    }
    xyzz_group_tree<TP>(acc, lane, tid, red);  // in the wave: DPP / ds_bpermute moves, no LDS (ecwave.h)
    if (lane == 0 && job < njobs) out[unit] = acc;
}
__global__ void k_pv_msm_combine(size_t njobs, int parts, const xyzz* __restrict__ partial, xyzz* __restrict__ out) {
    const size_t job = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (job >= njobs) return;
    xyzz acc = partial[job * parts];
    for (int p = 1; p < parts; p++) {
        xyzz o = partial[job * parts + p];
        xyzz_add(acc, o);
    }
    out[job] = acc;
}

__global__ void k_pv_final(size_t nproofs, int nm, const fqm* __restrict__ a, const fqm* __restrict__ b, uint8_t* __restrict__ out_be /* [N][64] */) {
    const size_t pr = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pr >= nproofs) return;
    fq_to_be(out_be + 64 * pr, fqm_to_fq(a[pr * nm]));
    fq_to_be(out_be + 64 * pr + 32, fqm_to_fq(b[pr * nm]));
}

// ====================================================================================================================
// Bulletproofs+ : scalar side of the BATCH prover `RangeProofWIP::prove` x N
// (/root/reference/src/proofs/range_proof_wip.rs:101-274, weighted_inner_product.rs:47-268), same shape as above.
// Tape of a proof (reference draw order, range_proof_wip.rs:150, weighted_inner_product.rs:106, 134, 213-219):
//   alpha, (d_L, d_R) per round, r, s, delta, eta
#define TW_ALPHA 0
#define TW_DL(j) (1 + 2 * (j))
#define TW_DR(j) (2 + 2 * (j))
#define TW_R(k) (1 + 2 * (k))
#define TW_S(k) (2 + 2 * (k))
#define TW_DELTA(k) (3 + 2 * (k))
#define TW_ETA(k) (4 + 2 * (k))

// A = alpha h + sum_{bit} g_i - sum_{!bit} h_i (range_proof_wip.rs:150-165): one job per proof, nm + W entries
__global__ __launch_bounds__(PV_T) void k_pw_init(int nm, int c, int W, const uint8_t* __restrict__ tapes, size_t tape_len,
                                                  const uint8_t* __restrict__ bits, pv_ent* __restrict__ ent, size_t stride) {
    const size_t pr = blockIdx.x;
    pv_ent* eA = ent + pr * stride;
    for (int i = threadIdx.x; i < nm; i += (int)blockDim.x) {
        const int bit = bits[pr * nm + i];
        pv_ent u;
        u.slot = (uint32_t)((bit ? i : nm + i) * W);
        u.d = bit ? 1 : -1;
        eA[i] = u;
    }
    if (threadIdx.x == 0) pv_digits(pv_be(tapes + pr * tape_len + 32 * TW_ALPHA), c, W, 2 * nm + 1, eA + nm);
}

// y^(i+1), the vectors of the weighted inner-product argument a = aL - z, b = aR + d_i y^(nm-i) + z (:254-260, d_i =
// 2^(i mod n) z^(2(i/n+1)) :186-199), alpha_hat = alpha + y^(nm+1) sum_j z^(2(j+1)) gamma_j (:205-207, 262), cG = cH = 1, y^-1
__global__ __launch_bounds__(PV_T) void k_pw_setup(int n_bits, int m, const uint8_t* __restrict__ tapes, size_t tape_len,
                                                   const uint8_t* __restrict__ yz_be, const uint8_t* __restrict__ blindings_be,
                                                   const uint8_t* __restrict__ bits, fqm* __restrict__ yp, fqm* __restrict__ a, fqm* __restrict__ b,
                                                   fqm* __restrict__ cG, fqm* __restrict__ cH, fqm* __restrict__ alpha_hat, fqm* __restrict__ yinv) {
    const size_t pr = blockIdx.x;
    const int nm = n_bits * m;
    const fqm y = pv_be(yz_be + 64 * pr), z = pv_be(yz_be + 64 * pr + 32), ONE = fqm_one(), zsq = fqm_mul(z, z);
    for (int i = threadIdx.x; i < nm; i += (int)blockDim.x) {
        yp[pr * nm + i] = pv_pow(y, (uint32_t)i + 1u);
        const fqm aL = bits[pr * nm + i] ? ONE : fqm_zero(), aR = fqm_sub(aL, ONE);
        const fqm d = fqm_mul(pv_pow(zsq, 1u + (uint32_t)(i / n_bits)), pv_two_pow(i % n_bits));
        a[pr * nm + i] = fqm_sub(aL, z);
        b[pr * nm + i] = fqm_add(aR, fqm_add(fqm_mul(d, pv_pow(y, (uint32_t)(nm - i))), z));
        cG[pr * nm + i] = ONE;
        cH[pr * nm + i] = ONE;
    }
    if (threadIdx.x == 0) {
        fqm ipb = fqm_zero(), zp = zsq;
        for (int j = 0; j < m; j++) {
            ipb = fqm_add(ipb, fqm_mul(pv_be(blindings_be + 32 * (pr * m + j)), zp));
            zp = fqm_mul(zp, zsq);
        }
        alpha_hat[pr] = fqm_add(pv_be(tapes + pr * tape_len + 32 * TW_ALPHA), fqm_mul(ipb, pv_pow(y, (uint32_t)nm + 1u)));
        yinv[pr] = fqm_inv_var(y);  // (y = 0 is refused by the host before this launch)
    }
}

// One round of WeightedInnerProdArg::prove over the ORIGINAL generators (n = current length, h = n / 2, round j):
// c_L = <a_L, b_R>_y, c_R = <y^h a_R, b_L>_y (:97-101, 128-130) and the entries of
//   L = <y^-h a_L, G_R> + <b_R, H_L> + c_L g + d_L h (:103-126),  R = <y^h a_R, G_L> + <b_L, H_R> + c_R g + d_R h (:131-154)
__global__ __launch_bounds__(PV_T) void k_pw_round(int nm, int n, int j, int c, int W, const uint8_t* __restrict__ tapes, size_t tape_len,
                                                   const fqm* __restrict__ yp, const fqm* __restrict__ yinv, const fqm* __restrict__ a,
                                                   const fqm* __restrict__ b, const fqm* __restrict__ cG, const fqm* __restrict__ cH,
                                                   pv_ent* __restrict__ ent, size_t stride) {
    __shared__ fqm red[PV_T];
    __shared__ fqm s_yninv;
    const size_t pr = blockIdx.x;
    const int h = n / 2;
    const fqm* A = a + pr * nm;
    const fqm* B = b + pr * nm;
    const fqm* Y = yp + pr * nm;
    const fqm yn = Y[h - 1];
    if (threadIdx.x == 0) s_yninv = pv_pow(yinv[pr], (uint32_t)h);
    fqm cL = fqm_zero(), cR = fqm_zero();
    for (int i = threadIdx.x; i < h; i += (int)blockDim.x) {
        cL = fqm_add(cL, fqm_mul(fqm_mul(A[i], B[h + i]), Y[i]));
        cR = fqm_add(cR, fqm_mul(fqm_mul(fqm_mul(yn, A[h + i]), B[i]), Y[i]));
    }
    cL = pv_block_sum(cL, red);
    cR = pv_block_sum(cR, red);
    const fqm yninv = s_yninv;
    pv_ent* eL = ent + (2 * pr) * stride;  // (nm + 2) W entries each
    pv_ent* eR = ent + (2 * pr + 1) * stride;
    for (int t = threadIdx.x; t < nm; t += (int)blockDim.x) {
        const int r = t & (n - 1);
        const fqm g = cG[pr * nm + t], hh = cH[pr * nm + t];
        if (r >= h) {
            pv_digits(fqm_mul(fqm_mul(yninv, A[r - h]), g), c, W, t, eL + (size_t)t * W);   // <y^-h a_L, G_R>
            pv_digits(fqm_mul(B[r - h], hh), c, W, nm + t, eR + (size_t)t * W);             // <b_L, H_R>
        } else {
            pv_digits(fqm_mul(B[h + r], hh), c, W, nm + t, eL + (size_t)t * W);             // <b_R, H_L>
            pv_digits(fqm_mul(fqm_mul(yn, A[h + r]), g), c, W, t, eR + (size_t)t * W);      // <y^h a_R, G_L>
        }
    }
    if (threadIdx.x == 0) {
        const uint8_t* tp = tapes + pr * tape_len;
        pv_digits(cL, c, W, 2 * nm, eL + (size_t)nm * W);
        pv_digits(pv_be(tp + 32 * (size_t)TW_DL(j)), c, W, 2 * nm + 1, eL + (size_t)(nm + 1) * W);
        pv_digits(cR, c, W, 2 * nm, eR + (size_t)nm * W);
        pv_digits(pv_be(tp + 32 * (size_t)TW_DR(j)), c, W, 2 * nm + 1, eR + (size_t)(nm + 1) * W);
    }
}

// a' = x a_L + x^-1 y^h a_R, b' = x b_R + x^-1 b_L (:164-179), alpha += x^2 d_L + x^-2 d_R (:182-185); the base folds
// G_hat = x^-1 G_L + (x y^-h) G_R, H_hat = x H_L + x^-1 H_R (:187-204) become updates of the coefficient vectors
__global__ __launch_bounds__(PV_T) void k_pw_fold(int nm, int n, int j, const uint8_t* __restrict__ tapes, size_t tape_len,
                                                  const uint8_t* __restrict__ xj_be, const fqm* __restrict__ yp, const fqm* __restrict__ yinv,
                                                  fqm* __restrict__ a, fqm* __restrict__ b, fqm* __restrict__ cG, fqm* __restrict__ cH,
                                                  fqm* __restrict__ alpha_hat) {
    __shared__ fqm xs[3];
    const size_t pr = blockIdx.x;
    const int h = n / 2;
    if (threadIdx.x == 0) {
        const fqm x = pv_be(xj_be + 32 * pr), xi = fqm_inv_var(x);  // (x = 0 is refused by the host before this launch)
        xs[0] = x;
        xs[1] = xi;
        xs[2] = fqm_mul(x, pv_pow(yinv[pr], (uint32_t)h));
        const uint8_t* tp = tapes + pr * tape_len;
        const fqm xsq = fqm_mul(x, x), xisq = fqm_mul(xi, xi);
        alpha_hat[pr] = fqm_add(alpha_hat[pr], fqm_add(fqm_mul(xsq, pv_be(tp + 32 * (size_t)TW_DL(j))), fqm_mul(xisq, pv_be(tp + 32 * (size_t)TW_DR(j)))));
    }
    __syncthreads();
    const fqm x = xs[0], xi = xs[1], xyn = xs[2], yn = yp[pr * nm + h - 1];
    fqm* A = a + pr * nm;
    fqm* B = b + pr * nm;
    for (int i = threadIdx.x; i < h; i += (int)blockDim.x) {  // lane i reads and writes only its own pair of slots
        const fqm an = fqm_add(fqm_mul(A[i], x), fqm_mul(fqm_mul(yn, A[h + i]), xi));
        const fqm bn = fqm_add(fqm_mul(B[h + i], x), fqm_mul(B[i], xi));
        A[i] = an;
        B[i] = bn;
    }
    for (int t = threadIdx.x; t < nm; t += (int)blockDim.x) {
        const bool lower = (t & (n - 1)) < h;
        cG[pr * nm + t] = fqm_mul(cG[pr * nm + t], lower ? xi : xyn);
        cH[pr * nm + t] = fqm_mul(cH[pr * nm + t], lower ? x : xi);
    }
}

// A' = r G_1 + s H_1 + (r y b + s y a) g + delta h (:222-232) and B' = r y s g + eta h (:235-239), G_1 / H_1 through the
// coefficient vectors: (2 nm + 2) W and 2 W entries
__global__ __launch_bounds__(PV_T) void k_pw_last(int nm, int k, int c, int W, const uint8_t* __restrict__ tapes, size_t tape_len,
                                                  const uint8_t* __restrict__ yz_be, const fqm* __restrict__ a, const fqm* __restrict__ b,
                                                  const fqm* __restrict__ cG, const fqm* __restrict__ cH, pv_ent* __restrict__ ent, size_t stride) {
    const size_t pr = blockIdx.x;
    const uint8_t* tp = tapes + pr * tape_len;
    const fqm r = pv_be(tp + 32 * (size_t)TW_R(k)), sv = pv_be(tp + 32 * (size_t)TW_S(k));
    pv_ent* eA = ent + (2 * pr) * stride;
    pv_ent* eB = ent + (2 * pr + 1) * stride;
    for (int t = threadIdx.x; t < nm; t += (int)blockDim.x) {
        pv_digits(fqm_mul(r, cG[pr * nm + t]), c, W, t, eA + (size_t)t * W);
        pv_digits(fqm_mul(sv, cH[pr * nm + t]), c, W, nm + t, eA + (size_t)(nm + t) * W);
    }
    if (threadIdx.x == 0) {
        const fqm y = pv_be(yz_be + 64 * pr), a0 = a[pr * nm], b0 = b[pr * nm];
        const fqm sG = fqm_add(fqm_mul(fqm_mul(a0, sv), y), fqm_mul(fqm_mul(b0, r), y));
        pv_digits(sG, c, W, 2 * nm, eA + (size_t)(2 * nm) * W);
        pv_digits(pv_be(tp + 32 * (size_t)TW_DELTA(k)), c, W, 2 * nm + 1, eA + (size_t)(2 * nm + 1) * W);
        pv_digits(fqm_mul(y, fqm_mul(r, sv)), c, W, 2 * nm, eB);
        pv_digits(pv_be(tp + 32 * (size_t)TW_ETA(k)), c, W, 2 * nm + 1, eB + W);
    }
}

// r' = r + a e, s' = s + b e, delta' = eta + delta e + alpha e^2 (:245-256) -> [N][96] big-endian
__global__ void k_pw_finish(size_t nproofs, int nm, int k, const uint8_t* __restrict__ tapes, size_t tape_len, const uint8_t* __restrict__ e_be,
                            const fqm* __restrict__ a, const fqm* __restrict__ b, const fqm* __restrict__ alpha_hat, uint8_t* __restrict__ out_be) {
    const size_t pr = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pr >= nproofs) return;
    const uint8_t* tp = tapes + pr * tape_len;
    const fqm e = pv_be(e_be + 32 * pr);
    const fqm r = pv_be(tp + 32 * (size_t)TW_R(k)), sv = pv_be(tp + 32 * (size_t)TW_S(k));
    const fqm delta = pv_be(tp + 32 * (size_t)TW_DELTA(k)), eta = pv_be(tp + 32 * (size_t)TW_ETA(k));
    fq_to_be(out_be + 96 * pr, fqm_to_fq(fqm_add(r, fqm_mul(a[pr * nm], e))));
    fq_to_be(out_be + 96 * pr + 32, fqm_to_fq(fqm_add(sv, fqm_mul(b[pr * nm], e))));
    fq_to_be(out_be + 96 * pr + 64, fqm_to_fq(fqm_add(eta, fqm_add(fqm_mul(delta, e), fqm_mul(alpha_hat[pr], fqm_mul(e, e))))));
}

// ---- launchers -----------------------------------------------------------------------------------------------------
hipError_t launch_pv_init(size_t nproofs, int nm, int c, int W, const uint8_t* tapes, size_t tape_len, const uint8_t* bits, fqm* SL, fqm* SR,
                          pv_ent* ent, size_t stride, hipStream_t st) {
    hipLaunchKernelGGL(k_pv_init, dim3((unsigned)nproofs), dim3(pv_lanes(nm)), 0, st, nm, c, W, tapes, tape_len, bits, SL, SR, ent, stride);
    return hipGetLastError();
}
hipError_t launch_pv_term(size_t nproofs, int c, int W, int row, const uint8_t* scalars_be, pv_ent* ent, size_t stride, hipStream_t st) {
    hipLaunchKernelGGL(k_pv_term, dim3((unsigned)((nproofs + 63) / 64)), dim3(64), 0, st, nproofs, c, W, row, scalars_be, ent, stride);
    return hipGetLastError();
}
hipError_t launch_pv_poly(size_t nproofs, int n_bits, int m, int c, int W, const uint8_t* tapes, size_t tape_len, const uint8_t* yz_be,
                          const uint8_t* bits, const fqm* SL, const fqm* SR, fqm* yi, fqm* zpow, fqm* yinv, pv_ent* ent, size_t stride,
                          hipStream_t st) {
    hipLaunchKernelGGL(k_pv_poly, dim3((unsigned)nproofs), dim3(pv_lanes(n_bits * m)), 0, st, n_bits, m, c, W, tapes, tape_len, yz_be, bits, SL, SR, yi, zpow, yinv,
                       ent, stride);
    return hipGetLastError();
}
hipError_t launch_pv_lr(size_t nproofs, int n_bits, int m, const uint8_t* tapes, size_t tape_len, const uint8_t* yz_be, const uint8_t* x_be,
                        const uint8_t* blindings_be, const uint8_t* bits, const fqm* SL, const fqm* SR, const fqm* yi, const fqm* zpow,
                        const fqm* yinv, fqm* a, fqm* b, fqm* cG, fqm* cH, uint8_t* out_be, hipStream_t st) {
    hipLaunchKernelGGL(k_pv_lr, dim3((unsigned)nproofs), dim3(pv_lanes(n_bits * m)), 0, st, n_bits, m, tapes, tape_len, yz_be, x_be, blindings_be, bits, SL, SR, yi,
                       zpow, yinv, a, b, cG, cH, out_be);
    return hipGetLastError();
}
hipError_t launch_pv_round(size_t nproofs, int nm, int n, int c, int W, const uint8_t* xu_be, const fqm* a, const fqm* b, const fqm* cG,
                           const fqm* cH, pv_ent* ent, size_t stride, hipStream_t st) {
    hipLaunchKernelGGL(k_pv_round, dim3((unsigned)nproofs), dim3(pv_lanes(nm)), 0, st, nm, n, c, W, xu_be, a, b, cG, cH, ent, stride);
    return hipGetLastError();
}
// njobs jobs of cnt0 (even) / cnt1 (odd) entries -> out[njobs]; lanes per job (and, past 256 lanes, workgroups per job) so that
// the launch has about four waves per SIMD.  `partial` holds njobs * PV_MSM_MAXPARTS sums.
hipError_t launch_pv_msm(size_t njobs, const affine_mem* table, int c, const pv_ent* ent, size_t stride, int cnt0, int cnt1, xyzz* partial,
                         xyzz* out, hipStream_t st) {
    if (njobs == 0) return hipSuccess;
    if (cnt0 < 1 || cnt1 < 1 || (size_t)cnt0 > stride || (size_t)cnt1 > stride) return hipErrorInvalidValue;
    const int cmax = cnt0 > cnt1 ? cnt0 : cnt1;
    int tp = 256, parts = 1;
    while (tp > 4 && (njobs * (size_t)tp > ((size_t)1 << 18) || tp * 4 > cmax)) tp >>= 1;
    if (tp == 256)
        while (parts < PV_MSM_MAXPARTS && njobs * (size_t)(256 * parts * 2) <= ((size_t)1 << 18) && 256 * parts * 2 * 8 <= cmax) parts *= 2;
    const size_t units = njobs * (size_t)parts;
    const unsigned blocks = (unsigned)((units + (256 / tp) - 1) / (256 / tp));
    xyzz* dst = parts > 1 ? partial : out;
#define PV_MSM(TP) hipLaunchKernelGGL((k_pv_msm<TP>), dim3(blocks), dim3(256), 0, st, njobs, table, c, ent, stride, cnt0, cnt1, parts, dst)
    if (tp == 256)
        PV_MSM(256);
    else if (tp == 128)
        PV_MSM(128);
    else if (tp == 64)
        PV_MSM(64);
    else if (tp == 32)
        PV_MSM(32);
    else if (tp == 16)
        PV_MSM(16);
    else if (tp == 8)
        PV_MSM(8);
    else
        PV_MSM(4);
#undef PV_MSM
    if (parts > 1) hipLaunchKernelGGL(k_pv_msm_combine, dim3((unsigned)((njobs + 63) / 64)), dim3(64), 0, st, njobs, parts, partial, out);
    return hipGetLastError();
}
hipError_t launch_pv_fold(size_t nproofs, int nm, int n, const uint8_t* xj_be, fqm* a, fqm* b, fqm* cG, fqm* cH, hipStream_t st) {
    hipLaunchKernelGGL(k_pv_fold, dim3((unsigned)nproofs), dim3(pv_lanes(nm)), 0, st, nm, n, xj_be, a, b, cG, cH);
    return hipGetLastError();
}
hipError_t launch_pv_final(size_t nproofs, int nm, const fqm* a, const fqm* b, uint8_t* out_be, hipStream_t st) {
    hipLaunchKernelGGL(k_pv_final, dim3((unsigned)((nproofs + 63) / 64)), dim3(64), 0, st, nproofs, nm, a, b, out_be);
    return hipGetLastError();
}

hipError_t launch_pw_init(size_t nproofs, int nm, int c, int W, const uint8_t* tapes, size_t tape_len, const uint8_t* bits, pv_ent* ent,
                          size_t stride, hipStream_t st) {
    hipLaunchKernelGGL(k_pw_init, dim3((unsigned)nproofs), dim3(pv_lanes(nm)), 0, st, nm, c, W, tapes, tape_len, bits, ent, stride);
    return hipGetLastError();
}
hipError_t launch_pw_setup(size_t nproofs, int n_bits, int m, const uint8_t* tapes, size_t tape_len, const uint8_t* yz_be,
                           const uint8_t* blindings_be, const uint8_t* bits, fqm* yp, fqm* a, fqm* b, fqm* cG, fqm* cH, fqm* alpha_hat, fqm* yinv,
                           hipStream_t st) {
    hipLaunchKernelGGL(k_pw_setup, dim3((unsigned)nproofs), dim3(pv_lanes(n_bits * m)), 0, st, n_bits, m, tapes, tape_len, yz_be, blindings_be, bits, yp, a,
                       b, cG, cH, alpha_hat, yinv);
    return hipGetLastError();
}
hipError_t launch_pw_round(size_t nproofs, int nm, int n, int j, int c, int W, const uint8_t* tapes, size_t tape_len, const fqm* yp, const fqm* yinv,
                           const fqm* a, const fqm* b, const fqm* cG, const fqm* cH, pv_ent* ent, size_t stride, hipStream_t st) {
    hipLaunchKernelGGL(k_pw_round, dim3((unsigned)nproofs), dim3(pv_lanes(nm)), 0, st, nm, n, j, c, W, tapes, tape_len, yp, yinv, a, b, cG, cH, ent, stride);
    return hipGetLastError();
}
hipError_t launch_pw_fold(size_t nproofs, int nm, int n, int j, const uint8_t* tapes, size_t tape_len, const uint8_t* xj_be, const fqm* yp,
                          const fqm* yinv, fqm* a, fqm* b, fqm* cG, fqm* cH, fqm* alpha_hat, hipStream_t st) {
    hipLaunchKernelGGL(k_pw_fold, dim3((unsigned)nproofs), dim3(pv_lanes(nm)), 0, st, nm, n, j, tapes, tape_len, xj_be, yp, yinv, a, b, cG, cH, alpha_hat);
    return hipGetLastError();
}
hipError_t launch_pw_last(size_t nproofs, int nm, int k, int c, int W, const uint8_t* tapes, size_t tape_len, const uint8_t* yz_be, const fqm* a,
                          const fqm* b, const fqm* cG, const fqm* cH, pv_ent* ent, size_t stride, hipStream_t st) {
    hipLaunchKernelGGL(k_pw_last, dim3((unsigned)nproofs), dim3(pv_lanes(nm)), 0, st, nm, k, c, W, tapes, tape_len, yz_be, a, b, cG, cH, ent, stride);
    return hipGetLastError();
}
hipError_t launch_pw_finish(size_t nproofs, int nm, int k, const uint8_t* tapes, size_t tape_len, const uint8_t* e_be, const fqm* a, const fqm* b,
                            const fqm* alpha_hat, uint8_t* out_be, hipStream_t st) {
    hipLaunchKernelGGL(k_pw_finish, dim3((unsigned)((nproofs + 63) / 64)), dim3(64), 0, st, nproofs, nm, k, tapes, tape_len, e_be, a, b, alpha_hat, out_be);
    return hipGetLastError();
}

}  // namespace bp
