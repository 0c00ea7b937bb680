// Variable-base scalar multiplication k * P for one point per lane (host + device).
//
// Used for the proof-specific terms of the verifier MSM (S, L_j, R_j, V_j, T1, T2 with 256-bit
// scalars: /root/reference/src/proofs/range_proof.rs:727-774), which cannot come from a table.
//   * GLV: k = k1 + k2*lambda with |k1|,|k2| < 2^129 and lambda*(x,y) = (beta*x, y), so one chain of
//     128 doublings serves both halves (instead of 256);
//   * fixed 4-bit signed windows, multiples 1P..8P kept as XYZZ in a caller-provided 8-entry slice
//     (lane-private global scratch on the GPU);
//   * uniform control flow: every lane does 4 doublings + 2 additions per window, so a wavefront
//     never serialises on "+P / -P / nothing" branches as a NAF ladder would.
#pragma once
#include "ec.h"
#include "fq.h"

namespace bp {

#define ECMUL_NT 8      // table entries per point
#define ECMUL_WINS 33   // 4-bit windows covering 129-bit magnitudes plus the signed-digit carry

BP_HD fp fp_beta() {
    fp b = {{0x119501eeu, 0x09cb6143u, 0x1d626570u, 0x0092ea25u, 0x034e99cfu, 0x03cf561au, 0x1c41b991u, 0x056caf80u, 0x007ae96au}};
    return b;
}

// carry-in bit per window for the signed recoding of a < 2^129 magnitude (digit = nibble + carry_in,
// minus 16 and carry out if > 8)
BP_HD uint64_t ecmul_carry_mask(const fq& k) {
    uint64_t mask = 0;
    uint32_t carry = 0;
    for (int w = 0; w < ECMUL_WINS - 1; w++) {
        uint32_t v = ((k.v[w >> 3] >> ((w & 7) * 4)) & 15u) + carry;
        carry = v > 8u ? 1u : 0u;
        mask |= (uint64_t)carry << (w + 1);
    }
    return mask;
}
BP_HD int ecmul_digit(const fq& k, uint64_t cmask, int w) {
    uint32_t v = ((k.v[w >> 3] >> ((w & 7) * 4)) & 15u) + (uint32_t)((cmask >> w) & 1u);
    return v > 8u ? (int)v - 16 : (int)v;
}

BP_HD xyzz ec_mul_glv(const affine& P, const fq& k, xyzz* tab) {
    if (affine_is_inf(P)) return xyzz_inf();
    fq k1, k2;
    bool n1, n2;
    fq_glv_split(k, k1, n1, k2, n2);
    xyzz t = xyzz_from_affine(P);
    tab[0] = t;
    t = xyzz_dbl_affine(P);
    tab[1] = t;
    for (int i = 2; i < ECMUL_NT; i++) {
        xyzz_madd(t, P);
        tab[i] = t;
    }
    const uint64_t c1 = ecmul_carry_mask(k1), c2 = ecmul_carry_mask(k2);
    const fp beta = fp_beta();
    xyzz acc = xyzz_inf();
    for (int w = ECMUL_WINS - 1; w >= 0; w--) {
        if (w != ECMUL_WINS - 1) {
            acc = xyzz_dbl(acc);
            acc = xyzz_dbl(acc);
            acc = xyzz_dbl(acc);
            acc = xyzz_dbl(acc);
        }
        int d1 = ecmul_digit(k1, c1, w);
        if (d1 != 0) {
            xyzz e = tab[(d1 < 0 ? -d1 : d1) - 1];
            if ((d1 < 0) != n1) e.Y = fp_neg(e.Y);
            xyzz_add(acc, e);
        }
        int d2 = ecmul_digit(k2, c2, w);
        if (d2 != 0) {
            xyzz e = tab[(d2 < 0 ? -d2 : d2) - 1];
            e.X = fp_mul(e.X, beta);
            if ((d2 < 0) != n2) e.Y = fp_neg(e.Y);
            xyzz_add(acc, e);
        }
    }
    return acc;
}

// One slice of k * P for latency-bound callers (few points, every lane of the wave free): slice = half + 2 * part,
// half selects the GLV component (k1 on P, k2 on lambda P), part one of `nparts` window ranges (part p: windows
// ec_part_lo(p) .. ec_part_lo(p + 1) - 1 on the base 2^(4 ec_part_lo(p)) * P_half).  The 2 nparts slices of a point run in
// as many lanes and are summed by the caller's lane tree.  nparts = 2 (windows 0..16 | 17..32): the longest lane does 68 +
// 60 doublings and 16 additions instead of 128 doublings and 66 additions.  The doublings on the critical path stay 128
// whatever nparts is (a part's base is that many doublings away); more parts move them from the ladder, which needs the
// scalar, into the table, which does not - nparts = 4: 100 doublings for the last table, 32 + 9 additions per ladder.
BP_HD int ec_part_lo(int part, int nparts) { return (part * ECMUL_WINS + nparts - 1) / nparts; }  // (2 parts: 0, 17, 33)
// The part of a slice that does not depend on the scalar: tab[i] = (i + 1) * base, base = 2^(4 w_lo) * P_half.  The batch
// verifier builds these tables beside its scalar kernels (k_bpv_chain), so that only ec_slice_ladder is left once the
// scalars exist.  P must not be the point at infinity.
BP_HD void ec_slice_table(const affine& P, int slice, xyzz* tab, int nparts = 2) {
    const int half = slice & 1, part = slice >> 1;
    affine Ph = P;
    if (half) Ph.x = fp_mul(P.x, fp_beta());
    xyzz base = xyzz_from_affine(Ph);
    if (part) base = xyzz_dbl_n(base, 4 * ec_part_lo(part, nparts));
    tab[0] = base;
    xyzz t = xyzz_dbl(base);
    tab[1] = t;
    for (int i = 2; i < ECMUL_NT; i++) {
        xyzz_add(t, base);
        tab[i] = t;
    }
}
// the slice's windows over a table made by ec_slice_table for the same (point, slice, nparts)
BP_HD xyzz ec_slice_ladder(const fq& k, int slice, const xyzz* tab, int nparts = 2) {
    fq k1, k2;
    bool n1, n2;
    fq_glv_split(k, k1, n1, k2, n2);
    const int half = slice & 1, part = slice >> 1;
    const fq kk = half ? k2 : k1;
    const bool neg = half ? n2 : n1;
    const int w_lo = ec_part_lo(part, nparts), w_hi = ec_part_lo(part + 1, nparts);
    const uint64_t cm = ecmul_carry_mask(kk);
    xyzz acc = xyzz_inf();
    for (int w = w_hi - 1; w >= w_lo; w--) {
        if (w != w_hi - 1) acc = xyzz_dbl_n(acc, 4);
        const int d = ecmul_digit(kk, cm, w);
        if (d != 0) {
            xyzz e = tab[(d < 0 ? -d : d) - 1];
            if ((d < 0) != neg) e.Y = fp_neg(e.Y);
            xyzz_add(acc, e);
        }
    }
    return acc;
}
BP_HD xyzz ec_mul_glv_slice(const affine& P, const fq& k, int slice, xyzz* tab, int nparts = 2) {
    if (affine_is_inf(P)) return xyzz_inf();
    ec_slice_table(P, slice, tab, nparts);
    return ec_slice_ladder(k, slice, tab, nparts);
}

// Interleaved (Straus) form for several points in ONE lane: sum_j k_j * P_j with a single chain of
// 128 doublings shared by all of them (each extra point costs only its 66 window additions).  Used by
// the batched verifier when a lane owns more than one proof-specific point.
//   loadP(j) -> affine, loadK(j) -> fq   for j < cnt <= PPL
//   tab : PPL * ECMUL_SLOTS lane-private XYZZ-sized slots (table construction: XYZZ multiples + prefix products),
//   atab: PPL * ECMUL_NT lane-private affine_mem entries (64 bytes each, one cache line: the finished tables),
//   dig : PPL * ECMUL_DIGB lane-private bytes, WINDOW-major: dig[w * 2 PPL + 2 j + half].
// The doubling chain runs through Jacobian coordinates (xyzz_dbl_n: 2M + 5S per doubling).
// The multiples 2P..8P of all the lane's points are brought to AFFINE form with ONE inversion (Montgomery's
// trick over up to 7 * PPL values of ZZZ; prefix products in the two spare slots of every point), so each of
// the ~62 window additions per point is a mixed addition (8M + 2S) instead of a full one (12M + 2S):
// 7 multiplications per entry + 270 for the inversion buy ~250 per point.
// The window loop is software-pipelined: all 2 cnt digits of a window come with one load issued a window ahead, and
// every table entry is fetched while the previous addition runs (the first one of a window during the four doublings):
// the lane-private tables of a full launch (hundreds of MB) live in HBM, and at two waves per SIMD a dependent
// digit -> entry -> addition sequence left the SIMD idle for a memory round trip per addition.
#define ECMUL_DIGB 72   // per point: 2 x 33 signed digits (k1, k2), padded
#define ECMUL_SLOTS 10  // per point: 8 table entries + 2 slots (8 field elements) of prefix products
template <int PPL, class LoadP, class LoadK>
BP_HD xyzz ec_mul_glv_multi(LoadP loadP, LoadK loadK, int cnt, xyzz* tab, affine_mem* atab, int8_t* dig) {
    constexpr int DW = 2 * PPL;  // digit bytes per window
    uint32_t valid = 0;
    fp run = fp_one();
    for (int j = 0; j < cnt; j++) {
        affine P = loadP(j);
        if (affine_is_inf(P)) {
            for (int w = 0; w < ECMUL_WINS; w++) dig[w * DW + 2 * j] = dig[w * DW + 2 * j + 1] = 0;
            continue;
        }
        valid |= 1u << j;
        xyzz* tj = tab + j * ECMUL_SLOTS;
        fp* pref = reinterpret_cast<fp*>(tj + ECMUL_NT);
        atab[j * ECMUL_NT] = affine_store(P);
        xyzz t = xyzz_dbl_affine(P);
        tj[1] = t;
        pref[1] = run;
        run = fp_mul(run, t.ZZZ);
        for (int i = 2; i < ECMUL_NT; i++) {
            xyzz_madd(t, P);
            tj[i] = t;
            pref[i] = run;
            run = fp_mul(run, t.ZZZ);
        }
        fq k1, k2;
        bool n1, n2;
        fq_glv_split(loadK(j), k1, n1, k2, n2);
        const uint64_t c1 = ecmul_carry_mask(k1), c2 = ecmul_carry_mask(k2);
        for (int w = 0; w < ECMUL_WINS; w++) {
            int d1 = ecmul_digit(k1, c1, w), d2 = ecmul_digit(k2, c2, w);
            dig[w * DW + 2 * j] = (int8_t)(n1 ? -d1 : d1);
            dig[w * DW + 2 * j + 1] = (int8_t)(n2 ? -d2 : d2);
        }
    }
    for (int j = cnt; j < PPL; j++)
        for (int w = 0; w < ECMUL_WINS; w++) dig[w * DW + 2 * j] = dig[w * DW + 2 * j + 1] = 0;
    if (valid) {
        fp inv = fp_inv_var(run);
        for (int j = cnt - 1; j >= 0; j--) {
            if (!((valid >> j) & 1)) continue;
            xyzz* tj = tab + j * ECMUL_SLOTS;
            const fp* pref = reinterpret_cast<const fp*>(tj + ECMUL_NT);
            for (int i = ECMUL_NT - 1; i >= 1; i--) {
                xyzz e = tj[i];
                fp zi3 = fp_mul(inv, pref[i]);  // 1 / ZZZ_i
                inv = fp_mul(inv, e.ZZZ);
                fp zi = fp_mul(e.ZZ, zi3);      // ZZ / ZZZ = 1 / Z
                affine a;
                a.x = fp_mul(e.X, fp_sqr(zi));
                a.y = fp_mul(e.Y, zi3);
                atab[j * ECMUL_NT + i] = affine_store(a);
            }
        }
    }
    const fp beta = fp_beta();
    // the 2 PPL digits of window w as packed bytes (one aligned load: the lane's digit block is 72 PPL bytes)
    auto load_digits = [&](int w, uint32_t dw[(DW + 3) / 4]) {
        if (DW >= 4) {
            const uint32_t* q = reinterpret_cast<const uint32_t*>(dig + w * DW);
            for (int i = 0; i < DW / 4; i++) dw[i] = q[i];
        } else {
            dw[0] = *reinterpret_cast<const uint16_t*>(dig + w * DW);
        }
    };
    auto digit_of = [&](const uint32_t dw[(DW + 3) / 4], int t) { return (int)(int8_t)(dw[t >> 2] >> (8 * (t & 3))); };
    // (a zero digit fetches entry 0 and ignores it: no branch around the load, nothing uninitialised)
    auto fetch = [&](int d, int t) { return affine_mem_ld(atab + (t >> 1) * ECMUL_NT + (d == 0 ? 0 : (d < 0 ? -d : d) - 1)); };
    const int nt = 2 * cnt;  // even: the k1 / k2 halves of cnt points
    uint32_t dcur[(DW + 3) / 4], dnext[(DW + 3) / 4];
    load_digits(ECMUL_WINS - 1, dcur);
    for (int i = 0; i < (DW + 3) / 4; i++) dnext[i] = 0;
    xyzz acc = xyzz_inf();
    // Two additions per trip through two entry buffers: A holds an even term (k1 half: the point itself), B an odd one (k2 half:
    // x times beta).  Every fetch is unconditional and outside any branch, and nothing is copied between buffers, so each
    // addition starts with the NEXT entry's four loads in flight and waits only for its own (exact vmcnt); the first entry
    // of the next window is fetched during the last addition of this one and stays in flight during the four doublings.
    // (The one-buffer form `en = em; if (t + 1 < nt) fetch(en)` compiled to a wait for the first piece of the new entry
    // right after its loads were issued - a memory round trip in front of every addition at two waves per SIMD.)
    int d0 = digit_of(dcur, 0);
    affine_mem A = fetch(d0, 0), B;
    for (int w = ECMUL_WINS - 1; w >= 0; w--) {
        if (w > 0) load_digits(w - 1, dnext);  // in flight during this window
        if (w != ECMUL_WINS - 1) acc = xyzz_dbl_n(acc, 4);
        for (int t = 0; t < nt; t += 2) {
            const int d1 = digit_of(dcur, t + 1);
            B = fetch(d1, t + 1);
            if (d0 != 0) {
                affine e = affine_load(A);
                xyzz_madd_s(acc, e, d0 < 0);
            }
            const bool more = t + 2 < nt;
            d0 = more ? digit_of(dcur, t + 2) : digit_of(dnext, 0);  // w == 0: dnext is stale, the entry is never used
            A = fetch(d0, more ? t + 2 : 0);
            if (d1 != 0) {
                affine e = affine_load(B);
                e.x = fp_mul(e.x, beta);
                xyzz_madd_s(acc, e, d1 < 0);
            }
        }
        for (int i = 0; i < (DW + 3) / 4; i++) dcur[i] = dnext[i];
    }
    return acc;
}

}  // namespace bp
