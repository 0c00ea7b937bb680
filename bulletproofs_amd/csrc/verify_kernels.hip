// Batched single-MSM verifier for Bulletproofs range proofs: a device-side restatement of
// `RangeProof::aggregated_verify` (/root/reference/src/proofs/range_proof.rs:537-782), one proof per
// group of lanes, thousands of proofs per launch.
//
//   k_bpv_decompress : one lane per proof point.  33-byte SEC1 -> affine limbs (curv
//                      `Point::from_bytes`), commitments ingested with the on-curve check.
//   k_bpv_scalars    : one lane per proof.  Fiat-Shamir challenges y, z, x, x_u, c, x_j
//                      (range_proof.rs:565-594, 650-668) with SHA-256 on device, the two fixed-base
//                      multiplications y*G and x_u*G (:571, :588) from the generator table, then the
//                      whole scalar vector of :670-740 in Montgomery arithmetic mod q, recoded to
//                      signed window digits for the table lookups.
//   k_bpv_msm        : TP lanes per proof.  The MSM of :742-774: fixed generators via table gathers
//                      (no doublings), proof-specific points (S, L_j, R_j, V_j, T1, T2) via NAF
//                      double-and-add, lane-tree reduction, and the comparison with A (:777).
//
// Per-proof verdicts follow `Result<(), Errors>`: 0 = Ok, 3 = RangeProofError, 255 = the proof bytes
// would not deserialize in the reference (bad point / scalar >= q / zero challenge).
#include <hip/hip_runtime.h>

#include "ec.h"
#include "fq.h"
#include "kernels.h"
#include "sha256.h"

namespace bp {

// ------------------------------------------------------------------------------------------------
__global__ void k_bpv_decompress(const uint8_t* __restrict__ proofs, size_t proof_len, const uint8_t* __restrict__ coms_xy, int k, int m,
                                 size_t nproofs, affine_mem* __restrict__ pts, int* __restrict__ bad) {
    const int npc = 4 + 2 * k;      // compressed points in a proof
    const int NPT = npc + m;        // + commitments
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nproofs * (size_t)NPT) return;
    size_t pr = gid / NPT;
    int j = (int)(gid % NPT);
    affine a;
    bool ok;
    if (j < npc) {
        // layout: A S T1 T2 | tau_x miu tx | L[k] | R[k] | a b   (oracle/encoding.py)
        size_t off = (j < 4) ? (size_t)33 * j : (size_t)(4 * 33 + 96) + (size_t)33 * (j - 4);
        const uint8_t* src = proofs + pr * proof_len + off;
        uint8_t b[33];
        for (int i = 0; i < 33; i++) b[i] = src[i];
        ok = affine_decompress(a, b);
        if (!ok) a = affine_inf();
    } else {
        const uint8_t* src = coms_xy + (pr * (size_t)m + (j - npc)) * 64;
        uint8_t b[64];
        for (int i = 0; i < 64; i++) b[i] = src[i];
        affine_mem mm = affine_mem_from_be64(b);
        ok = affine_mem_valid(mm, false);  // infinity is not a valid commitment encoding
        a = ok ? affine_load(mm) : affine_inf();
    }
    pts[pr * NPT + j] = affine_store(a);
    if (!ok) atomicOr(&bad[pr], 1);
}

// ------------------------------------------------------------------------------------------------
// transcript helpers (curv DigestExt: chain_points -> 65-byte uncompressed, chain_bigint -> minimal BE)
__device__ __forceinline__ void tr_point(sha256_ctx& c, const affine& p) {
    uint8_t b[65];
    b[0] = 4;
    affine_to_be64(b + 1, p);
    sha256_update(c, b, 65);
}
__device__ __forceinline__ void tr_bigint32(sha256_ctx& c, const uint8_t* be32) {
    int i = 0;
    while (i < 31 && be32[i] == 0) i++;  // zero -> single 0x00 byte
    sha256_update(c, be32 + i, 32 - i);
}
// result_scalar: first H(state || be32(i)) (i = 0, 1, ..) whose leading 32 bytes are < q.
__device__ __forceinline__ fq tr_result_scalar(const sha256_ctx& c) {
    for (uint32_t i = 0;; i++) {
        sha256_ctx t = c;
        uint8_t ctr[4] = {(uint8_t)(i >> 24), (uint8_t)(i >> 16), (uint8_t)(i >> 8), (uint8_t)i};
        sha256_update(t, ctr, 4);
        uint8_t d[32];
        sha256_final(t, d);
        if (fq_be_is_canonical(d)) return fq_from_be_reduce(d);
    }
}

// k * G from the generator table (G is fixed point index pG): W gathers + mixed adds, normalised.
__device__ __forceinline__ affine fixed_mul_table(const affine_mem* __restrict__ table, int pG, int c, int W, const fq& k) {
    const int NB = 1 << (c - 1);
    // recode (same signed-digit rule as recode_signed in msm_kernels.hip)
    uint32_t carry = 0;
    const uint32_t half = 1u << (c - 1), mask = (1u << c) - 1;
    xyzz acc = xyzz_inf();
    for (int w = 0; w < W; w++) {
        int bit = w * c;
        uint32_t v = 0;
        if (bit < 256) {
            int limb = bit >> 5, sh = bit & 31;
            uint64_t two = k.v[limb];
            if (limb + 1 < 8) two |= (uint64_t)k.v[limb + 1] << 32;
            v = (uint32_t)(two >> sh) & mask;
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
        if (d != 0) {
            affine e = affine_load(table[((size_t)pG * W + w) * NB + ((d < 0 ? -d : d) - 1)]);
            if (d < 0) e.y = fp_neg(e.y);
            xyzz_madd(acc, e);
        }
    }
    return xyzz_to_affine(acc);
}

__device__ __forceinline__ void store_digits(const fq& k, int c, int W, int16_t* __restrict__ out) {
    uint32_t carry = 0;
    const uint32_t half = 1u << (c - 1), mask = (1u << c) - 1;
    for (int w = 0; w < W; w++) {
        int bit = w * c;
        uint32_t v = 0;
        if (bit < 256) {
            int limb = bit >> 5, sh = bit & 31;
            uint64_t two = k.v[limb];
            if (limb + 1 < 8) two |= (uint64_t)k.v[limb + 1] << 32;
            v = (uint32_t)(two >> sh) & mask;
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
        out[w] = (int16_t)d;
    }
}

#define BPV_MAXK 16

// One lane per proof.  Outputs:
//   digF  [proof][(2nm+2) * W]  signed digits of the fixed-generator scalars (g_vec, h_vec, G, H)
//   varS  [proof][NV]           plain scalars of S, L_j, R_j, V_j, T1, T2 (NV = 3 + 2k + m)
//   sbuf  [proof][nm]           scratch (Montgomery s_i, range_proof.rs:670-680)
__global__ __launch_bounds__(64) void k_bpv_scalars(const uint8_t* __restrict__ proofs, size_t proof_len, int n_bits, int m, int k, size_t nproofs,
                                                    const affine_mem* __restrict__ pts, const affine_mem* __restrict__ gens /* g,h,G,H */,
                                                    const affine_mem* __restrict__ table, int c, int W, int16_t* __restrict__ digF,
                                                    fq* __restrict__ varS, fq* __restrict__ sbuf, int* __restrict__ bad) {
    size_t pr = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pr >= nproofs) return;
    if (bad[pr]) return;
    const int nm = n_bits * m;
    const int NPT = 4 + 2 * k + m;
    const int NV = 3 + 2 * k + m;
    const int pG = 2 * nm, pH = 2 * nm + 1;
    const affine_mem* P = pts + pr * NPT;
    const uint8_t* pf = proofs + pr * proof_len;
    const uint8_t* sc = pf + 4 * 33;                         // tau_x, miu, tx
    const uint8_t* ab = pf + 4 * 33 + 96 + (size_t)66 * k;   // a_tag, b_tag
    // scalars in the proof must be canonical (< q), as curv's Scalar/BigInt deserialisation yields
    bool okc = fq_be_is_canonical(sc) && fq_be_is_canonical(sc + 32) && fq_be_is_canonical(sc + 64) && fq_be_is_canonical(ab) &&
               fq_be_is_canonical(ab + 32);
    if (!okc) {
        atomicOr(&bad[pr], 1);
        return;
    }
    const affine A = affine_load(P[0]), S = affine_load(P[1]), T1 = affine_load(P[2]), T2 = affine_load(P[3]);
    const affine Gp = affine_load(gens[pG]), Hp = affine_load(gens[pH]);

    sha256_ctx h;
    // y = H(A, S)                                                     range_proof.rs:565-567
    sha256_init(h);
    tr_point(h, A);
    tr_point(h, S);
    fq y = tr_result_scalar(h);
    // yG = y * G ; z = H(yG)                                          :570-572
    affine yG = fixed_mul_table(table, pG, c, W, y);
    if (affine_is_inf(yG)) {  // y == 0: the reference would panic at mod_inv(...).unwrap() (:569)
        atomicOr(&bad[pr], 1);
        return;
    }
    sha256_init(h);
    tr_point(h, yG);
    fq z = tr_result_scalar(h);
    // x = H(T1, T2, G, H)                                             :576-578
    sha256_init(h);
    tr_point(h, T1);
    tr_point(h, T2);
    tr_point(h, Gp);
    tr_point(h, Hp);
    fq x = tr_result_scalar(h);
    // x_u = H(tau_x, miu, tx) ; ux = x_u * G                          :581-588
    sha256_init(h);
    tr_bigint32(h, sc);
    tr_bigint32(h, sc + 32);
    tr_bigint32(h, sc + 64);
    fq xu = tr_result_scalar(h);
    affine ux = fixed_mul_table(table, pG, c, W, xu);
    if (affine_is_inf(ux)) {  // x_u == 0: the identity has no 65-byte encoding to hash (:656)
        atomicOr(&bad[pr], 1);
        return;
    }
    // c = H(A, S, T1, T2, G, H)                                       :591-593
    sha256_init(h);
    tr_point(h, A);
    tr_point(h, S);
    tr_point(h, T1);
    tr_point(h, T2);
    tr_point(h, Gp);
    tr_point(h, Hp);
    fq cv = tr_result_scalar(h);

    // Montgomery domain from here on
    const fq ONE = fq_mont_one();
    fq y_m = fq_to_mont(y), z_m = fq_to_mont(z), x_m = fq_to_mont(x), xu_m = fq_to_mont(xu), c_m = fq_to_mont(cv);
    fq taux_m = fq_to_mont(fq_from_be_reduce(sc)), miu_m = fq_to_mont(fq_from_be_reduce(sc + 32)), tx_m = fq_to_mont(fq_from_be_reduce(sc + 64));
    fq a_m = fq_to_mont(fq_from_be_reduce(ab)), b_m = fq_to_mont(fq_from_be_reduce(ab + 32));

    // x_j = H(L_j, R_j, ux)                                           :650-656
    fq xj[BPV_MAXK], pref[BPV_MAXK + 1];
    fq run = y_m;  // batch inversion of (y, x_0 .. x_{k-1})
    pref[0] = run;
    bool zero_ch = fq_is_zero(y);
    for (int j = 0; j < k; j++) {
        sha256_init(h);
        tr_point(h, affine_load(P[4 + j]));
        tr_point(h, affine_load(P[4 + k + j]));
        tr_point(h, ux);
        fq xv = tr_result_scalar(h);
        zero_ch = zero_ch || fq_is_zero(xv);
        xj[j] = fq_to_mont(xv);
        run = fq_mont_mul(run, xj[j]);
        pref[j + 1] = run;
    }
    if (zero_ch) {  // `.invert().unwrap()` would panic (:658)
        atomicOr(&bad[pr], 1);
        return;
    }
    fq inv = fq_mont_inv(run);
    fq xinv[BPV_MAXK];
    for (int j = k - 1; j >= 0; j--) {
        xinv[j] = fq_mont_mul(inv, pref[j]);
        inv = fq_mont_mul(inv, xj[j]);
    }
    fq yinv_m = inv;

    fq* s = sbuf + pr * (size_t)nm;
    fq allinv = ONE;
    fq* vs = varS + pr * (size_t)NV;
    for (int j = 0; j < k; j++) {
        allinv = fq_mont_mul(allinv, xinv[j]);
        fq xsq = fq_mont_mul(xj[j], xj[j]);
        fq xinvsq = fq_mont_mul(xinv[j], xinv[j]);
        vs[1 + j] = fq_from_mont(fq_neg(xsq));          // L_j : -x_j^2        :665, 730
        vs[1 + k + j] = fq_from_mont(fq_neg(xinvsq));   // R_j : -x_j^-2       :666, 731
        xj[j] = xsq;                                    // keep x_j^2 for the s recursion
    }
    // s_0 = prod x_j^-1 ; s_i = s_{i - 2^lg(i)} * x_{k-1-lg(i)}^2     :670-680
    s[0] = allinv;
    for (int i = 1; i < nm; i++) {
        int lg = 31 - __clz(i);
        s[i] = fq_mont_mul(s[i - (1 << lg)], xj[(k - 1) - lg]);
    }
    // g_i : a*s_i + z ; h_i : y^-i (b*s_i^-1 - d_i) - z  with s_i^-1 = s_{nm-1-i}   :682-705
    int16_t* dg = digF + pr * (size_t)(2 * nm + 2) * W;
    fq ypow = ONE, yinvpow = ONE, sum_y = fq_zero(), sum_d = fq_zero();
    fq zsq = fq_mont_mul(z_m, z_m);
    fq zpow = zsq;  // z^(2+j)
    fq d_i = zpow;
    for (int i = 0; i < nm; i++) {
        if (i > 0 && (i % n_bits) == 0) {
            zpow = fq_mont_mul(zpow, z_m);
            d_i = zpow;
        }
        fq gs = fq_add(fq_mont_mul(a_m, s[i]), z_m);
        fq bs = fq_mont_mul(b_m, s[nm - 1 - i]);
        fq hs = fq_sub(fq_mont_mul(yinvpow, fq_sub(bs, d_i)), z_m);
        store_digits(fq_from_mont(gs), c, W, dg + (size_t)i * W);
        store_digits(fq_from_mont(hs), c, W, dg + (size_t)(nm + i) * W);
        sum_y = fq_add(sum_y, ypow);
        sum_d = fq_add(sum_d, d_i);
        ypow = fq_mont_mul(ypow, y_m);
        yinvpow = fq_mont_mul(yinvpow, yinv_m);
        d_i = fq_add(d_i, d_i);
    }
    // delta = (z - z^2) sum y^i - z sum d_i                          :639-642
    fq delta = fq_sub(fq_mont_mul(fq_sub(z_m, zsq), sum_y), fq_mont_mul(sum_d, z_m));
    // G : x_u (ab - tx) + c (delta - tx) ; H : miu - c tau_x         :708-723
    fq abm = fq_mont_mul(a_m, b_m);
    fq sG = fq_add(fq_mont_mul(xu_m, fq_sub(abm, tx_m)), fq_mont_mul(c_m, fq_sub(delta, tx_m)));
    fq sH = fq_sub(miu_m, fq_mont_mul(c_m, taux_m));
    store_digits(fq_from_mont(sG), c, W, dg + (size_t)pG * W);
    store_digits(fq_from_mont(sH), c, W, dg + (size_t)pH * W);
    // S : -x ; V_j : c z^(2+j) ; T1 : c x ; T2 : c x^2               :727-740
    vs[0] = fq_from_mont(fq_neg(x_m));
    fq zp = zsq;
    for (int j = 0; j < m; j++) {
        vs[1 + 2 * k + j] = fq_from_mont(fq_mont_mul(c_m, zp));
        zp = fq_mont_mul(zp, z_m);
    }
    fq cx = fq_mont_mul(c_m, x_m);
    vs[1 + 2 * k + m] = fq_from_mont(cx);
    vs[2 + 2 * k + m] = fq_from_mont(fq_mont_mul(cx, x_m));
}

// ------------------------------------------------------------------------------------------------
// k * P for one variable point: fixed 4-bit signed windows (digits in [-8, 8], 65 windows), with the
// multiples 1P..8P kept as XYZZ in a lane-private slice of a global scratch buffer.  Control flow is
// uniform across the wave (4 doublings + 1 addition per window for every lane): a NAF
// double-and-add would make every lane wait on both the +P and the -P branch at almost every bit.
#define VAR_WBITS 4
#define VAR_NT 8  // table entries
__device__ __forceinline__ xyzz window_mul(const affine& P, const fq& k, xyzz* __restrict__ tab) {
    if (affine_is_inf(P)) return xyzz_inf();
    xyzz t = xyzz_from_affine(P);
    tab[0] = t;
    t = xyzz_dbl_affine(P);
    tab[1] = t;
    for (int i = 2; i < VAR_NT; i++) {
        xyzz_madd(t, P);
        tab[i] = t;
    }
    // signed digits, most significant window first; recode on the fly from the top:
    // d_w = bits(w) + carry_in(w) - 16*carry_out(w) needs the carries from below, so precompute them.
    // carry into window w+1 is 1 iff (bits(w) + carry(w)) > 8.
    uint32_t cmask[3] = {0, 0, 0};  // carry-in bit per window (65 windows)
    {
        uint32_t carry = 0;
        for (int w = 0; w < 64; w++) {
            uint32_t v = ((k.v[w >> 3] >> ((w & 7) * 4)) & 15u) + carry;
            carry = v > 8u ? 1u : 0u;
            if (carry) cmask[(w + 1) >> 5] |= 1u << ((w + 1) & 31);
        }
    }
    xyzz acc = xyzz_inf();
    for (int w = 64; w >= 0; w--) {
        if (w != 64) {
            acc = xyzz_dbl(acc);
            acc = xyzz_dbl(acc);
            acc = xyzz_dbl(acc);
            acc = xyzz_dbl(acc);
        }
        uint32_t bits = (w < 64) ? ((k.v[w >> 3] >> ((w & 7) * 4)) & 15u) : 0u;
        uint32_t v = bits + ((cmask[w >> 5] >> (w & 31)) & 1u);
        int d = v > 8u ? (int)v - 16 : (int)v;
        if (d != 0) {
            xyzz e = tab[(d < 0 ? -d : d) - 1];
            if (d < 0) e.Y = fp_neg(e.Y);
            xyzz_add(acc, e);
        }
    }
    return acc;
}

// TP lanes per proof; 256-thread workgroups.
template <int TP>
__global__ __launch_bounds__(256) void k_bpv_msm(int n_bits, int m, int k, size_t nproofs, const affine_mem* __restrict__ pts,
                                                 const affine_mem* __restrict__ table, int c, int W, const int16_t* __restrict__ digF,
                                                 const fq* __restrict__ varS, xyzz* __restrict__ vartab, const int* __restrict__ bad,
                                                 uint8_t* __restrict__ verdicts) {
    __shared__ xyzz red[256];
    const int tid = threadIdx.x;
    const int lane = tid % TP;
    const size_t pr = (size_t)blockIdx.x * (256 / TP) + tid / TP;
    const bool active = pr < nproofs && !(pr < nproofs && bad[pr]);
    const int nm = n_bits * m;
    const int NPT = 4 + 2 * k + m;
    const int NV = 3 + 2 * k + m;
    const int NB = 1 << (c - 1);
    xyzz acc = xyzz_inf();
    if (active) {
        const int nfw = (2 * nm + 2) * W;
        const int16_t* dg = digF + pr * (size_t)nfw;
        for (int idx = lane; idx < nfw; idx += TP) {
            int d = dg[idx];
            if (d != 0) {
                affine e = affine_load(table[(size_t)idx * NB + ((d < 0 ? -d : d) - 1)]);
                if (d < 0) e.y = fp_neg(e.y);
                xyzz_madd(acc, e);
            }
        }
        const affine_mem* P = pts + pr * NPT;
        const fq* vs = varS + pr * (size_t)NV;
        for (int v = lane; v < NV; v += TP) {
            // variable point order: S, L_0.., R_0.., V_0.., T1, T2  ->  index into pts
            int pi = (v == 0) ? 1 : (v < 1 + 2 * k + m) ? (3 + v) : (v == 1 + 2 * k + m ? 2 : 3);
            xyzz t = window_mul(affine_load(P[pi]), vs[v], vartab + ((size_t)blockIdx.x * 256 + tid) * VAR_NT);
            xyzz_add(acc, t);
        }
    }
    red[tid] = acc;
    __syncthreads();
    for (int off = TP / 2; off >= 1; off >>= 1) {
        if (lane < off) {
            xyzz v = red[tid];
            xyzz o = red[tid + off];
            xyzz_add(v, o);
            red[tid] = v;
        }
        __syncthreads();
    }
    if (lane == 0 && pr < nproofs) {
        uint8_t verdict;
        if (bad[pr]) {
            verdict = 255;
        } else {
            // lhs == A ?  (range_proof.rs:777) compared projectively: X = A.x*ZZ, Y = A.y*ZZZ
            xyzz r = red[tid];
            affine A = affine_load(pts[pr * NPT]);
            bool eq = !xyzz_is_inf(r) && fp_eq(r.X, fp_mul(A.x, r.ZZ)) && fp_eq(r.Y, fp_mul(A.y, r.ZZZ));
            verdict = eq ? 0 : 3;
        }
        verdicts[pr] = verdict;
    }
}

hipError_t launch_bpv_decompress(const uint8_t* proofs, size_t proof_len, const uint8_t* coms_xy, int k, int m, size_t nproofs, affine_mem* pts,
                                 int* bad, hipStream_t st) {
    size_t total = nproofs * (size_t)(4 + 2 * k + m);
    hipLaunchKernelGGL(k_bpv_decompress, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, proofs, proof_len, coms_xy, k, m, nproofs, pts,
                       bad);
    return hipGetLastError();
}
hipError_t launch_bpv_scalars(const uint8_t* proofs, size_t proof_len, int n_bits, int m, int k, size_t nproofs, const affine_mem* pts,
                              const affine_mem* gens, const affine_mem* table, int c, int W, int16_t* digF, fq* varS, fq* sbuf, int* bad,
                              hipStream_t st) {
    hipLaunchKernelGGL(k_bpv_scalars, dim3((unsigned)((nproofs + 63) / 64)), dim3(64), 0, st, proofs, proof_len, n_bits, m, k, nproofs, pts, gens,
                       table, c, W, digF, varS, sbuf, bad);
    return hipGetLastError();
}
size_t bpv_msm_vartab_bytes(int tp, size_t nproofs) {
    size_t blocks = (nproofs + (256 / tp) - 1) / (256 / tp);
    return blocks * 256 * VAR_NT * sizeof(xyzz);
}
hipError_t launch_bpv_msm(int tp, int n_bits, int m, int k, size_t nproofs, const affine_mem* pts, const affine_mem* table, int c, int W,
                          const int16_t* digF, const fq* varS, xyzz* vartab, const int* bad, uint8_t* verdicts, hipStream_t st) {
    unsigned blocks = (unsigned)((nproofs + (256 / tp) - 1) / (256 / tp));
    if (tp == 16)
        hipLaunchKernelGGL(k_bpv_msm<16>, dim3(blocks), dim3(256), 0, st, n_bits, m, k, nproofs, pts, table, c, W, digF, varS, vartab, bad, verdicts);
    else if (tp == 32)
        hipLaunchKernelGGL(k_bpv_msm<32>, dim3(blocks), dim3(256), 0, st, n_bits, m, k, nproofs, pts, table, c, W, digF, varS, vartab, bad, verdicts);
    else
        hipLaunchKernelGGL(k_bpv_msm<64>, dim3(blocks), dim3(256), 0, st, n_bits, m, k, nproofs, pts, table, c, W, digF, varS, vartab, bad, verdicts);
    return hipGetLastError();
}

}  // namespace bp
