// secp256k1 scalar field F_q (group order q) on 8 x 32-bit limbs, Montgomery form (R = 2^256).
//
// Replaces curv-kzen's `Scalar<Secp256k1>` / `BigInt::mod_{add,sub,mul,inv,pow}` arithmetic used to
// derive the verifier's scalar vector (/root/reference/src/proofs/range_proof.rs:565-740,
// inner_product.rs:276-317, range_proof_wip.rs:419-574).  q is not pseudo-Mersenne, so products use
// a CIOS Montgomery multiply (64 + 64 + 8 `v_mad_u64_u32`) as north_star prescribes.  `fq` values
// are plain integers in [0, q); `fqm` values are in the Montgomery domain.  Conversions are
// explicit so the canonical 32-byte big-endian scalars at the boundary are bit-identical to the
// reference's.
#pragma once
#include "fp.h"

namespace bp {

struct fq {
    uint32_t v[8];
};

// bits [bit, bit + 32) of the 256-bit value k[0..8) (zero above bit 255), 0 <= bit < 256.  The limb index is a run-time value:
// indexing k[limb] directly makes the compiler keep the eight words in scratch memory and every window digit two scratch loads
// (a 36-byte private segment in every digit kernel; k_pw_round spent 0.6 ms per launch there); sixteen selects keep them in
// registers.
BP_HD uint32_t fq_bits32(const uint32_t k[8], int bit) {
    const int limb = bit >> 5, sh = bit & 31;
    uint32_t lo = 0, hi = 0;
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll
#endif
    for (int i = 0; i < 8; i++) {
        lo = (limb == i) ? k[i] : lo;
        hi = (limb + 1 == i) ? k[i] : hi;
    }
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh);
}

#define BP_Q_LIMBS {0xd0364141u, 0xbfd25e8cu, 0xaf48a03bu, 0xbaaedce6u, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0xffffffffu}
#define BP_Q_INV32 0x5588b13fu /* -q^-1 mod 2^32 */

BP_HD uint32_t fq_q(int i) {
    const uint32_t q[8] = BP_Q_LIMBS;
    return q[i];
}
BP_HD fq fq_zero() {
    fq r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
}
BP_HD fq fq_from_u32(uint32_t x) {
    fq r = fq_zero();
    r.v[0] = x;
    return r;
}
BP_HD bool fq_is_zero(const fq& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
}
BP_HD bool fq_eq(const fq& a, const fq& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
}
// r = a - q; returns borrow (1 if a < q).
BP_HD uint32_t fq_sub_q(uint32_t r[8], const uint32_t a[8]) {
    const uint32_t q[8] = BP_Q_LIMBS;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (int64_t)a[i] - (int64_t)q[i];
        r[i] = (uint32_t)c;
        c >>= 32;
    }
    return (uint32_t)(c & 1);
}
// value = carry*2^256 + r < 2q  ->  r mod q
BP_HD void fq_final_reduce(uint32_t r[8], uint32_t carry) {
    uint32_t s[8];
    uint32_t borrow = fq_sub_q(s, r);
    if (carry | (borrow ^ 1u)) {
#pragma unroll
        for (int i = 0; i < 8; i++) r[i] = s[i];
    }
}
BP_HD fq fq_add(const fq& a, const fq& b) {
    fq r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.v[i] + b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    fq_final_reduce(r.v, (uint32_t)c);
    return r;
}
BP_HD fq fq_sub(const fq& a, const fq& b) {
    const uint32_t q[8] = BP_Q_LIMBS;
    fq r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (int64_t)a.v[i] - (int64_t)b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    if (c) {
        uint64_t d = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            d += (uint64_t)r.v[i] + q[i];
            r.v[i] = (uint32_t)d;
            d >>= 32;
        }
    }
    return r;
}
BP_HD fq fq_neg(const fq& a) { return fq_sub(fq_zero(), a); }

// Montgomery product a*b*R^-1 mod q (CIOS).  Inputs < q.
BP_HD fq fq_mont_mul(const fq& a, const fq& b) {
    const uint32_t q[8] = BP_Q_LIMBS;
    uint32_t t[10];
#pragma unroll
    for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c += (uint64_t)a.v[i] * b.v[j] + t[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[8] = (uint32_t)c;
        t[9] = (uint32_t)(c >> 32);
        uint32_t m = t[0] * BP_Q_INV32;
        c = (uint64_t)m * q[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            c += (uint64_t)m * q[j] + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[7] = (uint32_t)c;
        t[8] = t[9] + (uint32_t)(c >> 32);
    }
    fq r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    fq_final_reduce(r.v, t[8]);
    return r;
}

BP_HD fq fq_R2() {
    fq r = {{0x67d7d140u, 0x896cf214u, 0x0e7cf878u, 0x741496c2u, 0x5bcd07c6u, 0xe697f5e4u, 0x81c69bc5u, 0x9d671cd5u}};
    return r;
}
BP_HD fq fq_mont_one() {  // R mod q
    fq r = {{0x2fc9bebfu, 0x402da173u, 0x50b75fc4u, 0x45512319u, 0x00000001u, 0, 0, 0}};
    return r;
}
BP_HD fq fq_to_mont(const fq& a) { return fq_mont_mul(a, fq_R2()); }
BP_HD fq fq_from_mont(const fq& a) { return fq_mont_mul(a, fq_from_u32(1)); }

// Plain-domain product a*b mod q (two Montgomery multiplies); for the host protocol layer.
BP_HD fq fq_mul(const fq& a, const fq& b) { return fq_mont_mul(fq_mont_mul(a, b), fq_R2()); }

BP_HD fq fq_mont_sqr_n(fq a, int n) {
    for (int i = 0; i < n; i++) a = fq_mont_mul(a, a);
    return a;
}

// a^(q-2) in the Montgomery domain (input and output in Montgomery form). a != 0.
// q - 2 = (2^127 - 1) * 2^129 + low: the run of 127 ones by the usual 2^k - 1 chain (126 squarings, 10
// multiplications), the low 129 bits by a sliding window over the odd powers a^1..a^15 (129 squarings,
// 26 + 7 multiplications): 298 Montgomery products instead of the 447 of bit-by-bit square-and-multiply
// (this inversion is the serial chain that bounds k_bpv_consts).
BP_HD_NOINLINE fq fq_mont_inv(fq a) {
    const uint8_t nsqr[26] = {5, 3, 4, 4, 5, 2, 5, 6, 5, 4, 3, 6, 10, 4, 5, 4, 5, 6, 4, 5, 6, 10, 4, 9, 4, 1};
    const uint8_t odd[26] = {11, 5, 5, 7, 13, 3, 7, 13, 11, 13, 1, 5, 7, 7, 15, 15, 9, 11, 13, 3, 13, 13, 9, 9, 15, 1};
    fq tbl[8];  // a^1, a^3, ..., a^15
    tbl[0] = a;
    fq a2 = fq_mont_mul(a, a);
    for (int i = 1; i < 8; i++) tbl[i] = fq_mont_mul(tbl[i - 1], a2);
    fq x2 = tbl[1], x3 = tbl[3];  // a^(2^2-1), a^(2^3-1)
    fq x6 = fq_mont_mul(fq_mont_sqr_n(x3, 3), x3);
    fq x12 = fq_mont_mul(fq_mont_sqr_n(x6, 6), x6);
    fq x24 = fq_mont_mul(fq_mont_sqr_n(x12, 12), x12);
    fq x48 = fq_mont_mul(fq_mont_sqr_n(x24, 24), x24);
    fq x96 = fq_mont_mul(fq_mont_sqr_n(x48, 48), x48);
    fq x120 = fq_mont_mul(fq_mont_sqr_n(x96, 24), x24);
    fq x126 = fq_mont_mul(fq_mont_sqr_n(x120, 6), x6);
    fq acc = fq_mont_mul(fq_mont_mul(x126, x126), a);  // a^(2^127-1)
    (void)x2;
    for (int s = 0; s < 26; s++) {
        acc = fq_mont_sqr_n(acc, nsqr[s]);
        acc = fq_mont_mul(acc, tbl[odd[s] >> 1]);
    }
    return acc;
}
BP_HD fq fq_inv(const fq& a) { return fq_from_mont(fq_mont_inv(fq_to_mont(a))); }

// ---- reduced-radix Montgomery arithmetic mod q for the device-side scalar pipelines --------------
// `fqm`: 9 limbs of 29 bits, Montgomery domain with R = 2^261, value kept < 2q ("weakly reduced").
// The 8x32 CIOS product above compiles to ~500 VALU instructions (every partial product drags a 64-bit
// carry addition); here a whole column (<= 18 products < 2^58) accumulates in one 64-bit register pair:
// 81 + 81 v_mad_u64_u32, nine 32-bit multiplies for the quotient digits, one mask/shift per column.
// Since R = 32 * 2^256 > 32 q, inputs < 2q give (ab + mq)/R < q/8 + q: no conditional subtraction
// inside the product.
struct fqm {
    uint32_t v[9];
};
#define BP_FQM_Q                                                                                                        \
    { 0x10364141u, 0x1e92f466u, 0x12280eefu, 0x1db9cd5eu, 0x1fffebaau, 0x1fffffffu, 0x1fffffffu, 0x1fffffffu, 0x00ffffffu }
#define BP_FQM_2Q                                                                                                       \
    { 0x006c8282u, 0x1d25e8cdu, 0x04501ddfu, 0x1b739abdu, 0x1fffd755u, 0x1fffffffu, 0x1fffffffu, 0x1fffffffu, 0x01ffffffu }
#define BP_FQM_QINV 0x1588b13fu  // -q^-1 mod 2^29
#define BP_M29Q 0x1FFFFFFFu

#if defined(__HIP_DEVICE_COMPILE__)
#define BP_FQ_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define BP_FQ_OPAQUE(x) ((void)0)
#endif

BP_HD fqm fqm_one() {  // R mod q
    fqm r = {{0x1937d7e0u, 0x0da1732fu, 0x1afe2201u, 0x08c6542du, 0x00028aa2u, 0, 0, 0, 0}};
    return r;
}
BP_HD fqm fqm_zero() {
    fqm r = {{0, 0, 0, 0, 0, 0, 0, 0, 0}};
    return r;
}

// v >= K ? v - K : v   (limbs of v and K normalised: < 2^29, top limb free)
BP_HD void fqm_csub(uint32_t v[9], const uint32_t K[9]) {
    uint32_t t[9];
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int32_t d = (int32_t)v[i] - (int32_t)K[i] + borrow;
        t[i] = (uint32_t)d & BP_M29Q;
        borrow = d >> 29;
    }
    int32_t d = (int32_t)v[8] - (int32_t)K[8] + borrow;
    if (d >= 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = t[i];
        v[8] = (uint32_t)d;
    }
}

BP_HD fqm fqm_mul(fqm a, fqm b) {
    uint32_t q[9] = BP_FQM_Q;
    uint32_t qinv = BP_FQM_QINV;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        BP_FQ_OPAQUE(a.v[i]);
        BP_FQ_OPAQUE(b.v[i]);
        BP_FQ_OPAQUE(q[i]);  // keep the multiplier: q's limbs 2^29 - 1 would otherwise become 64-bit shift/subtract pairs
    }
    BP_FQ_OPAQUE(qinv);
    uint32_t m[9];
    fqm r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * q[k - i];
        m[k] = ((uint32_t)acc * qinv) & BP_M29Q;
        BP_FQ_OPAQUE(m[k]);
        acc += (uint64_t)m[k] * q[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) {
            acc += (uint64_t)a.v[i] * b.v[k - i];
            acc += (uint64_t)m[i] * q[k - i];
        }
        r.v[k - 9] = (uint32_t)acc & BP_M29Q;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}
BP_HD fqm fqm_sqr_n(fqm a, int n) {
    for (int i = 0; i < n; i++) a = fqm_mul(a, a);
    return a;
}

// a + b, a - b: inputs < 2q, result < 2q
BP_HD fqm fqm_add(const fqm& a, const fqm& b) {
    const uint32_t q2[9] = BP_FQM_2Q;
    fqm r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t t = a.v[i] + b.v[i] + c;
        r.v[i] = t & BP_M29Q;
        c = t >> 29;
    }
    r.v[8] = a.v[8] + b.v[8] + c;
    fqm_csub(r.v, q2);
    return r;
}
BP_HD fqm fqm_sub(const fqm& a, const fqm& b) {
    const uint32_t q2[9] = BP_FQM_2Q;
    fqm r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int32_t t = (int32_t)a.v[i] + (int32_t)q2[i] - (int32_t)b.v[i] + c;
        r.v[i] = (uint32_t)t & BP_M29Q;
        c = t >> 29;
    }
    r.v[8] = (uint32_t)((int32_t)a.v[8] + (int32_t)q2[8] - (int32_t)b.v[8] + c);
    fqm_csub(r.v, q2);
    return r;
}
BP_HD fqm fqm_neg(const fqm& a) { return fqm_sub(fqm_zero(), a); }

// plain fq (canonical, < q) -> Montgomery domain, and back to the canonical plain value
BP_HD fqm fqm_r2() {  // R^2 mod q: fqm_mul(plain limbs, R^2) = Montgomery form
    const fqm r2 = {{0x09f6ab4bu, 0x1f300d1eu, 0x1c0bd5d8u, 0x0c8ada8cu, 0x11cefa2bu, 0x08b79a0fu, 0x1e697f5eu, 0x00e34de2u, 0x009c7356u}};
    return r2;
}
BP_HD fqm fqm_from_fq(const fq& a) {
    const uint32_t* w = a.v;
    fqm t;
    t.v[0] = w[0] & BP_M29Q;
    t.v[1] = ((w[0] >> 29) | (w[1] << 3)) & BP_M29Q;
    t.v[2] = ((w[1] >> 26) | (w[2] << 6)) & BP_M29Q;
    t.v[3] = ((w[2] >> 23) | (w[3] << 9)) & BP_M29Q;
    t.v[4] = ((w[3] >> 20) | (w[4] << 12)) & BP_M29Q;
    t.v[5] = ((w[4] >> 17) | (w[5] << 15)) & BP_M29Q;
    t.v[6] = ((w[5] >> 14) | (w[6] << 18)) & BP_M29Q;
    t.v[7] = ((w[6] >> 11) | (w[7] << 21)) & BP_M29Q;
    t.v[8] = w[7] >> 8;
    return fqm_mul(t, fqm_r2());
}
// a / R mod q (Montgomery reduction alone: the m*q half of fqm_mul, 90 MADs instead of 171); result < q + 1
BP_HD fqm fqm_redc(fqm a) {
    uint32_t q[9] = BP_FQM_Q;
    uint32_t qinv = BP_FQM_QINV;
#pragma unroll
    for (int i = 0; i < 9; i++) BP_FQ_OPAQUE(q[i]);
    BP_FQ_OPAQUE(qinv);
    uint32_t m[9];
    fqm r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        acc += a.v[k];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * q[k - i];
        m[k] = ((uint32_t)acc * qinv) & BP_M29Q;
        BP_FQ_OPAQUE(m[k]);
        acc += (uint64_t)m[k] * q[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) acc += (uint64_t)m[i] * q[k - i];
        r.v[k - 9] = (uint32_t)acc & BP_M29Q;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}
BP_HD fq fqm_to_fq(const fqm& a) {
    fqm c = fqm_redc(a);  // a / R  < q + 1
    const uint32_t q[9] = BP_FQM_Q;
    fqm_csub(c.v, q);
    const uint32_t* v = c.v;
    fq s;
    s.v[0] = v[0] | (v[1] << 29);
    s.v[1] = (v[1] >> 3) | (v[2] << 26);
    s.v[2] = (v[2] >> 6) | (v[3] << 23);
    s.v[3] = (v[3] >> 9) | (v[4] << 20);
    s.v[4] = (v[4] >> 12) | (v[5] << 17);
    s.v[5] = (v[5] >> 15) | (v[6] << 14);
    s.v[6] = (v[6] >> 18) | (v[7] << 11);
    s.v[7] = (v[7] >> 21) | (v[8] << 8);
    return s;
}
// value == 0 mod q ?  (weakly reduced: 0 or q)
BP_HD bool fqm_is_zero(const fqm& a) {
    const uint32_t q[9] = BP_FQM_Q;
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        z |= a.v[i];
        e |= a.v[i] ^ q[i];
    }
    return z == 0 || e == 0;
}

// a^(q-2), same exponent schedule as fq_mont_inv
BP_HD_NOINLINE fqm fqm_inv(fqm a) {
    const uint8_t nsqr[26] = {5, 3, 4, 4, 5, 2, 5, 6, 5, 4, 3, 6, 10, 4, 5, 4, 5, 6, 4, 5, 6, 10, 4, 9, 4, 1};
    const uint8_t odd[26] = {11, 5, 5, 7, 13, 3, 7, 13, 11, 13, 1, 5, 7, 7, 15, 15, 9, 11, 13, 3, 13, 13, 9, 9, 15, 1};
    fqm tbl[8];
    tbl[0] = a;
    fqm a2 = fqm_mul(a, a);
    for (int i = 1; i < 8; i++) tbl[i] = fqm_mul(tbl[i - 1], a2);
    fqm x3 = tbl[3];
    fqm x6 = fqm_mul(fqm_sqr_n(x3, 3), x3);
    fqm x12 = fqm_mul(fqm_sqr_n(x6, 6), x6);
    fqm x24 = fqm_mul(fqm_sqr_n(x12, 12), x12);
    fqm x48 = fqm_mul(fqm_sqr_n(x24, 24), x24);
    fqm x96 = fqm_mul(fqm_sqr_n(x48, 48), x48);
    fqm x120 = fqm_mul(fqm_sqr_n(x96, 24), x24);
    fqm x126 = fqm_mul(fqm_sqr_n(x120, 6), x6);
    fqm acc = fqm_mul(fqm_mul(x126, x126), a);
    for (int s = 0; s < 26; s++) {
        acc = fqm_sqr_n(acc, nsqr[s]);
        acc = fqm_mul(acc, tbl[odd[s] >> 1]);
    }
    return acc;
}

// The same inverse by division steps (fp.h, s30_inv_words) - about a sixth of the instructions of the exponentiation
// above, data-dependent round counts (a wavefront runs the longest of its lanes: 19-22 rounds of 25 possible).
// In:  a = x R (weakly reduced).  x R -> canonical words -> (x R)^-1 -> times R^3 / R = x^-1 R.  0 -> 0.
struct s30_mod_q {
    static BP_HD int32_t limb(int i) {
        return i == 0 ? 0x10364141 : i == 1 ? 0x3f497a33 : i == 2 ? 0x348a03bb : i == 3 ? 0x2bb739ab : i == 4 ? 0x3ffffeba : (i < 8 ? 0x3fffffff : 0xffff);
    }
    static BP_HD uint32_t inv30() { return 0x2a774ec1u; }  // q^-1 mod 2^30
};
BP_HD fqm fqm_inv_var(fqm a) {
    const uint32_t q[9] = BP_FQM_Q;
    fqm_csub(a.v, q);  // < 2q -> canonical
    const uint32_t* v = a.v;
    uint32_t w[8], r[8];
    w[0] = v[0] | (v[1] << 29);
    w[1] = (v[1] >> 3) | (v[2] << 26);
    w[2] = (v[2] >> 6) | (v[3] << 23);
    w[3] = (v[3] >> 9) | (v[4] << 20);
    w[4] = (v[4] >> 12) | (v[5] << 17);
    w[5] = (v[5] >> 15) | (v[6] << 14);
    w[6] = (v[6] >> 18) | (v[7] << 11);
    w[7] = (v[7] >> 21) | (v[8] << 8);
    s30_inv_words<s30_mod_q>(w, r);
    fqm t;
    t.v[0] = r[0] & BP_M29Q;
    t.v[1] = ((r[0] >> 29) | (r[1] << 3)) & BP_M29Q;
    t.v[2] = ((r[1] >> 26) | (r[2] << 6)) & BP_M29Q;
    t.v[3] = ((r[2] >> 23) | (r[3] << 9)) & BP_M29Q;
    t.v[4] = ((r[3] >> 20) | (r[4] << 12)) & BP_M29Q;
    t.v[5] = ((r[4] >> 17) | (r[5] << 15)) & BP_M29Q;
    t.v[6] = ((r[5] >> 14) | (r[6] << 18)) & BP_M29Q;
    t.v[7] = ((r[6] >> 11) | (r[7] << 21)) & BP_M29Q;
    t.v[8] = r[7] >> 8;
    const fqm r3 = {{0x15617bd2u, 0x1b88b909u, 0x1be1f2e0u, 0x1449ba16u, 0x136481f2u, 0x1c742cb6u, 0x0366268fu, 0x0ef116dbu, 0x00c0060cu}};  // R^3 mod q
    return fqm_mul(t, r3);
}

// ---- GLV endomorphism split (secp256k1: lambda^3 = 1 mod q, phi(x, y) = (beta x, y) = lambda * (x, y)) ----
// k = k1 + k2 * lambda (mod q) with |k1|, |k2| < 2^128 (constants and method of libsecp256k1's
// scalar_split_lambda).  Halves the doubling chain of every variable-base multiplication.
// ((k * g + 2^383) >> 384), k and g 256-bit: the rounded quotient used for c1, c2 (< 2^128).
BP_HD fq fq_mul_shift384(const fq& k, const uint32_t g[8]) {
    uint32_t t[16];
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        c += (uint64_t)k.v[0] * g[j];
        t[j] = (uint32_t)c;
        c >>= 32;
    }
    t[8] = (uint32_t)c;
#pragma unroll
    for (int i = 1; i < 8; i++) {
        c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c += (uint64_t)k.v[i] * g[j] + t[i + j];
            t[i + j] = (uint32_t)c;
            c >>= 32;
        }
        t[i + 8] = (uint32_t)c;
    }
    // + 2^383 = bit 31 of limb 11, then take limbs 12..15
    c = (uint64_t)t[11] + 0x80000000u;
    c >>= 32;
    fq r = fq_zero();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        c += t[12 + i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}
// out[0 .. 4 + NB) = a[0..4) * b[0..NB): the 128-bit quotients times a lattice constant, plain integers
template <int NB>
BP_HD void fq_mul4(uint32_t* out, const uint32_t a[4], const uint32_t* b) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < NB; j++) {
        c += (uint64_t)a[0] * b[j];
        out[j] = (uint32_t)c;
        c >>= 32;
    }
    out[NB] = (uint32_t)c;
#pragma unroll
    for (int i = 1; i < 4; i++) {
        c = 0;
#pragma unroll
        for (int j = 0; j < NB; j++) {
            c += (uint64_t)a[i] * b[j] + out[i + j];
            out[i + j] = (uint32_t)c;
            c >>= 32;
        }
        out[i + NB] = (uint32_t)c;
    }
}
// k (plain, < q) -> magnitudes k1, k2 (< 2^129, plain) and their signs (true = negative).
// With the rounded quotients c1 = round(b2 k / q), c2 = round(-b1 k / q) (< 2^128) the halves are small INTEGERS,
//   k1 = k - c1 a1 - c2 a2,   k2 = -c1 b1 - c2 b2        (lattice basis (a1, b1), (a2, b2), a_i + b_i lambda = 0 mod q),
// so four 128 x 128(129)-bit products and a few subtractions give them directly - 68 multiplies, where three Montgomery
// multiplications mod q (k2 = c1 (-b1) + c2 (-b2), k1 = k - k2 lambda) took 400: k_glv_digits 168 -> 110 us on rocprofv3's average over
// the MSM bench's calls (about 80 -> 52 us per 2^20 scalars).
// Same values as that form on every input (tests/test_host_arith.py::test_fq_glv_split against a Python model of both).
BP_HD void fq_glv_split(const fq& k, fq& k1, bool& neg1, fq& k2, bool& neg2) {
    const uint32_t G1[8] = {0x45dbb031u, 0xe893209au, 0x71e8ca7fu, 0x3daa8a14u, 0x9284eb15u, 0xe86c90e4u, 0xa7d46bcdu, 0x3086d221u};
    const uint32_t G2[8] = {0x8ac47f71u, 0x1571b4aeu, 0x9df506c6u, 0x221208acu, 0x0abfe4c4u, 0x6f547fa9u, 0x010e8828u, 0xe4437ed6u};
    const uint32_t A1[4] = {0x9284eb15u, 0xe86c90e4u, 0xa7d46bcdu, 0x3086d221u};               // a1 = b2
    const uint32_t MB1[4] = {0x0abfe4c3u, 0x6f547fa9u, 0x010e8828u, 0xe4437ed6u};              // -b1
    const uint32_t A2[5] = {0x9d44cfd8u, 0x57c1108du, 0xa8e2f3f6u, 0x14ca50f7u, 0x00000001u};  // a2 (129 bits)
    const fq c1 = fq_mul_shift384(k, G1), c2 = fq_mul_shift384(k, G2);
    uint32_t t1[8], t2[9], u1[8], u2[8];
    fq_mul4<4>(t1, c1.v, A1);
    fq_mul4<5>(t2, c2.v, A2);
    fq_mul4<4>(u1, c1.v, MB1);
    fq_mul4<4>(u2, c2.v, A1);
    // r1 = k - t1 - t2, r2 = u1 - u2 as 9-limb two's complement numbers (|r| < 2^129)
    uint32_t r1[9], r2[9];
    int64_t b = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        b += (int64_t)(i < 8 ? k.v[i] : 0u) - (int64_t)(i < 8 ? t1[i] : 0u) - (int64_t)t2[i];
        r1[i] = (uint32_t)b;
        b >>= 32;
    }
    b = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        b += (int64_t)(i < 8 ? u1[i] : 0u) - (int64_t)(i < 8 ? u2[i] : 0u);
        r2[i] = (uint32_t)b;
        b >>= 32;
    }
    neg1 = (r1[8] >> 31) != 0;
    neg2 = (r2[8] >> 31) != 0;
    // magnitudes: conditional two's-complement negation, low 8 limbs (the ninth is a sign extension)
    uint64_t c = neg1 ? 1 : 0;
    const uint32_t m1 = neg1 ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)(r1[i] ^ m1);
        k1.v[i] = (uint32_t)c;
        c >>= 32;
    }
    c = neg2 ? 1 : 0;
    const uint32_t m2 = neg2 ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)(r2[i] ^ m2);
        k2.v[i] = (uint32_t)c;
        c >>= 32;
    }
}

// 32 big-endian bytes -> scalar reduced mod q (`Scalar::from(&BigInt)` for 256-bit inputs).
BP_HD fq fq_from_be_reduce(const uint8_t* b) {
    fq r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint8_t* p = b + 4 * (7 - i);
        r.v[i] = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
    }
    fq_final_reduce(r.v, 0);  // 2^256 < 2q, one conditional subtraction suffices
    return r;
}
// true iff the 32 bytes encode a value < q (curv `Scalar::from_bytes` acceptance, zero allowed).
BP_HD bool fq_be_is_canonical(const uint8_t* b) {
    uint32_t a[8], s[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint8_t* p = b + 4 * (7 - i);
        a[i] = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
    }
    return fq_sub_q(s, a) == 1;
}
BP_HD void fq_to_be(uint8_t* b, const fq& a) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint8_t* p = b + 4 * (7 - i);
        p[0] = (uint8_t)(a.v[i] >> 24);
        p[1] = (uint8_t)(a.v[i] >> 16);
        p[2] = (uint8_t)(a.v[i] >> 8);
        p[3] = (uint8_t)(a.v[i]);
    }
}

}  // namespace bp
