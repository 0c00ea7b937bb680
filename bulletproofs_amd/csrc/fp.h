// secp256k1 base field F_p, p = 2^256 - 2^32 - 977, on 8 x 32-bit limbs (little-endian limb order).
//
// Replaces the field arithmetic libsecp256k1 performs behind curv-kzen's `Point<Secp256k1>` operators
// (reference call sites: every `&Point * &Scalar` / `Point + Point`, e.g.
// /root/reference/src/proofs/range_proof.rs:773-774).  Written for CDNA4's 32-bit VALU: products are
// `v_mad_u64_u32` chains (acc64 += a32*b32), reduction uses the pseudo-Mersenne fold
// 2^256 = 2^32 + 977 (mod p) instead of a Montgomery pass: 9 extra multiplies instead of 72, and the
// values stay in the plain (non-Montgomery) domain so they are bit-identical to the canonical
// coordinates the reference serialises.  All results are fully reduced to [0, p).
//
// Every function is __host__ __device__ so the same code is unit-tested on the CPU (tests/) and
// used by the host-side protocol layer.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BP_HD __host__ __device__ __forceinline__
#define BP_HD_NOINLINE inline __host__ __device__ __noinline__
// fp_mul / fp_sqr are ~300 VALU instructions each; a point addition inlines 10-14 of them.  Calling
// them as real functions keeps kernels inside the instruction cache (override to measure).
#ifndef BP_FP_MUL_INLINE
#define BP_HD_MUL inline __host__ __device__ __noinline__
#else
#define BP_HD_MUL __host__ __device__ __forceinline__
#endif
#else
#define BP_HD inline
#define BP_HD_NOINLINE inline
#define BP_HD_MUL inline
#endif

namespace bp {

struct fp {
    uint32_t v[8];
};

// 2^256 - p = 2^32 + 977
#define BP_P_C0 977u

BP_HD fp fp_zero() {
    fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
}
BP_HD fp fp_one() {
    fp r = fp_zero();
    r.v[0] = 1;
    return r;
}
BP_HD bool fp_is_zero(const fp& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
}
BP_HD bool fp_eq(const fp& a, const fp& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
}

// r + (2^32 + 977) over 8 limbs; returns the carry out of limb 7.
BP_HD uint32_t fp_add_c(uint32_t s[8], const uint32_t r[8]) {
    uint64_t c = (uint64_t)r[0] + BP_P_C0;
    s[0] = (uint32_t)c;
    c >>= 32;
    c += (uint64_t)r[1] + 1u;
    s[1] = (uint32_t)c;
    c >>= 32;
#pragma unroll
    for (int i = 2; i < 8; i++) {
        c += r[i];
        s[i] = (uint32_t)c;
        c >>= 32;
    }
    return (uint32_t)c;
}

// Final conditional subtraction of p given value = carry*2^256 + r, value < 2p.
BP_HD void fp_final_reduce(uint32_t r[8], uint32_t carry) {
    uint32_t s[8];
    uint32_t c2 = fp_add_c(s, r);
    if (carry | c2) {
#pragma unroll
        for (int i = 0; i < 8; i++) r[i] = s[i];
    }
}

BP_HD fp fp_add(const fp& a, const fp& b) {
    fp r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.v[i] + b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    fp_final_reduce(r.v, (uint32_t)c);
    return r;
}

BP_HD fp fp_sub(const fp& a, const fp& b) {
    fp r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (int64_t)a.v[i] - (int64_t)b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;  // arithmetic shift: 0 or -1
    }
    if (c) {
        // a - b + 2^256 - (2^32 + 977) = a - b + p
        int64_t d = (int64_t)r.v[0] - (int64_t)BP_P_C0;
        r.v[0] = (uint32_t)d;
        d >>= 32;
        d += (int64_t)r.v[1] - 1;
        r.v[1] = (uint32_t)d;
        d >>= 32;
#pragma unroll
        for (int i = 2; i < 8; i++) {
            d += (int64_t)r.v[i];
            r.v[i] = (uint32_t)d;
            d >>= 32;
        }
    }
    return r;
}

BP_HD fp fp_neg(const fp& a) { return fp_sub(fp_zero(), a); }
BP_HD fp fp_dbl(const fp& a) { return fp_add(a, a); }

// Reduce a 512-bit product t[0..15] modulo p.
BP_HD fp fp_reduce512(const uint32_t t[16]) {
    // u = hi * 977 (9 limbs)
    uint32_t u[9];
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        c += (uint64_t)t[8 + j] * BP_P_C0;
        u[j] = (uint32_t)c;
        c >>= 32;
    }
    u[8] = (uint32_t)c;
    // r = lo + u + (hi << 32); limb 8 collects the overflow T (< 2^34)
    fp r;
    c = (uint64_t)t[0] + u[0];
    r.v[0] = (uint32_t)c;
    c >>= 32;
#pragma unroll
    for (int k = 1; k < 8; k++) {
        c += (uint64_t)t[k] + u[k];
        c += t[8 + k - 1];
        r.v[k] = (uint32_t)c;
        c >>= 32;
    }
    uint64_t T = c + (uint64_t)u[8] + t[15];
    // fold T * (2^32 + 977) back in
    uint64_t m = T * BP_P_C0;  // < 2^44
    c = (uint64_t)r.v[0] + (uint32_t)m;
    r.v[0] = (uint32_t)c;
    c >>= 32;
    c += (uint64_t)r.v[1] + (m >> 32) + (uint32_t)T;
    r.v[1] = (uint32_t)c;
    c >>= 32;
    c += (uint64_t)r.v[2] + (T >> 32);
    r.v[2] = (uint32_t)c;
    c >>= 32;
#pragma unroll
    for (int k = 3; k < 8; k++) {
        c += r.v[k];
        r.v[k] = (uint32_t)c;
        c >>= 32;
    }
    fp_final_reduce(r.v, (uint32_t)c);
    return r;
}

// 8x8 schoolbook product, operand scanning: 64 multiply-accumulates (the MAD32 unit of SURVEY.md §8d).
BP_HD void fp_mul_wide(uint32_t t[16], const fp& a, const fp& b) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        c += (uint64_t)a.v[0] * b.v[j];
        t[j] = (uint32_t)c;
        c >>= 32;
    }
    t[8] = (uint32_t)c;
#pragma unroll
    for (int i = 1; i < 8; i++) {
        c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c += (uint64_t)a.v[i] * b.v[j] + t[i + j];
            t[i + j] = (uint32_t)c;
            c >>= 32;
        }
        t[i + 8] = (uint32_t)c;
    }
}

BP_HD_MUL fp fp_mul(fp a, fp b) {
    uint32_t t[16];
    fp_mul_wide(t, a, b);
    return fp_reduce512(t);
}

// Squaring: 36 multiplies (off-diagonal products computed once and doubled).
BP_HD_MUL fp fp_sqr(fp a) {
    uint32_t t[16];
    // off-diagonal: sum_{i<j} a_i a_j 2^{32(i+j)}
    uint64_t c = 0;
    t[0] = 0;
#pragma unroll
    for (int j = 1; j < 8; j++) {
        c += (uint64_t)a.v[0] * a.v[j];
        t[j] = (uint32_t)c;
        c >>= 32;
    }
    t[8] = (uint32_t)c;
#pragma unroll
    for (int i = 1; i < 7; i++) {
        c = 0;
#pragma unroll
        for (int j = i + 1; j < 8; j++) {
            c += (uint64_t)a.v[i] * a.v[j] + t[i + j];
            t[i + j] = (uint32_t)c;
            c >>= 32;
        }
        t[i + 8] = (uint32_t)c;
    }
    t[15] = 0;
    // double and add the diagonal
    uint32_t top = 0;
#pragma unroll
    for (int k = 15; k > 0; k--) t[k] = (t[k] << 1) | (t[k - 1] >> 31);
    t[0] <<= 1;
    (void)top;
    c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t d = (uint64_t)a.v[i] * a.v[i];
        c += (uint64_t)t[2 * i] + (uint32_t)d;
        t[2 * i] = (uint32_t)c;
        c >>= 32;
        c += (uint64_t)t[2 * i + 1] + (d >> 32);
        t[2 * i + 1] = (uint32_t)c;
        c >>= 32;
    }
    return fp_reduce512(t);
}

BP_HD fp fp_sqr_n(fp a, int n) {
    for (int i = 0; i < n; i++) a = fp_sqr(a);
    return a;
}

// a^(p-2) by the standard secp256k1 addition chain (255 squarings + 15 multiplications).
BP_HD_NOINLINE fp fp_inv(fp a) {
    fp x2 = fp_mul(fp_sqr(a), a);
    fp x3 = fp_mul(fp_sqr(x2), a);
    fp x6 = fp_mul(fp_sqr_n(x3, 3), x3);
    fp x9 = fp_mul(fp_sqr_n(x6, 3), x3);
    fp x11 = fp_mul(fp_sqr_n(x9, 2), x2);
    fp x22 = fp_mul(fp_sqr_n(x11, 11), x11);
    fp x44 = fp_mul(fp_sqr_n(x22, 22), x22);
    fp x88 = fp_mul(fp_sqr_n(x44, 44), x44);
    fp x176 = fp_mul(fp_sqr_n(x88, 88), x88);
    fp x220 = fp_mul(fp_sqr_n(x176, 44), x44);
    fp x223 = fp_mul(fp_sqr_n(x220, 3), x3);
    fp t = fp_mul(fp_sqr_n(x223, 23), x22);
    t = fp_mul(fp_sqr_n(t, 5), a);
    t = fp_mul(fp_sqr_n(t, 3), x2);
    t = fp_mul(fp_sqr_n(t, 2), a);
    return t;
}

// a^((p+1)/4): a square root of a when a is a quadratic residue (p = 3 mod 4).
BP_HD_NOINLINE fp fp_sqrt_candidate(fp a) {
    fp x2 = fp_mul(fp_sqr(a), a);
    fp x3 = fp_mul(fp_sqr(x2), a);
    fp x6 = fp_mul(fp_sqr_n(x3, 3), x3);
    fp x9 = fp_mul(fp_sqr_n(x6, 3), x3);
    fp x11 = fp_mul(fp_sqr_n(x9, 2), x2);
    fp x22 = fp_mul(fp_sqr_n(x11, 11), x11);
    fp x44 = fp_mul(fp_sqr_n(x22, 22), x22);
    fp x88 = fp_mul(fp_sqr_n(x44, 44), x44);
    fp x176 = fp_mul(fp_sqr_n(x88, 88), x88);
    fp x220 = fp_mul(fp_sqr_n(x176, 44), x44);
    fp x223 = fp_mul(fp_sqr_n(x220, 3), x3);
    fp t = fp_mul(fp_sqr_n(x223, 23), x22);
    t = fp_mul(fp_sqr_n(t, 6), x2);
    t = fp_sqr_n(t, 2);
    return t;
}

// 32 big-endian bytes <-> limbs.  from_be does not reduce; callers check fp_is_canonical.
BP_HD fp fp_from_be(const uint8_t* b) {
    fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint8_t* q = b + 4 * (7 - i);
        r.v[i] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | (uint32_t)q[3];
    }
    return r;
}
BP_HD void fp_to_be(uint8_t* b, const fp& a) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint8_t* q = b + 4 * (7 - i);
        q[0] = (uint8_t)(a.v[i] >> 24);
        q[1] = (uint8_t)(a.v[i] >> 16);
        q[2] = (uint8_t)(a.v[i] >> 8);
        q[3] = (uint8_t)(a.v[i]);
    }
}
// value < p ?
BP_HD bool fp_is_canonical(const fp& a) {
    uint32_t s[8];
    return fp_add_c(s, a.v) == 0;
}

}  // namespace bp
