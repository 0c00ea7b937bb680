// secp256k1 point arithmetic (y^2 = x^3 + 7) in XYZZ coordinates: x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2.
//
// Replaces `Point<Secp256k1>` `+`, `-`, `*` of curv-kzen (call sites: every fold idiom of SURVEY.md
// §2.2, e.g. /root/reference/src/proofs/range_proof.rs:770-774).  XYZZ is chosen because the MSM
// hot loop is affine-into-accumulator additions: 8M + 2S (mixed) with no inversion until the very
// end.  The reference's generator recipe makes h_vec[i] == g_vec[i+n] (range_proof.rs:837-841), so
// equal and opposite operands DO occur inside one accumulator: every add handles P+P and P+(-P).
//
// Types: `affine` / `xyzz` are the in-register compute forms (fp = 9x29-bit limbs);
// `affine_mem` is the 64-byte canonical storage form used for every point array in HBM (tables,
// generators, proof points): one cache line per gather.
// Infinity markers: affine x = y = 0 (raw zero limbs; not a curve point since b = 7), xyzz ZZ = 0
// (raw).  They are only ever created by affine_inf()/xyzz_inf() or zeroed memory, never by
// arithmetic, so the tests are raw limb tests (no canonicalisation needed).
#pragma once
#include "fp.h"

namespace bp {

struct affine {
    fp x, y;
};
struct xyzz {
    fp X, Y, ZZ, ZZZ;
};
struct affine_mem {  // 64 bytes: x words then y words, canonical; all-zero = infinity
    fp_words x, y;
};

BP_HD bool fp_raw_is_zero(const fp& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= a.v[i];
    return o == 0;
}
BP_HD bool affine_is_inf(const affine& p) { return fp_raw_is_zero(p.x) && fp_raw_is_zero(p.y); }
BP_HD affine affine_inf() {
    affine r;
    r.x = fp_zero();
    r.y = fp_zero();
    return r;
}
BP_HD affine affine_neg(const affine& p) {
    if (affine_is_inf(p)) return p;
    affine r;
    r.x = p.x;
    r.y = fp_neg(p.y);
    return r;
}
BP_HD bool affine_eq(const affine& a, const affine& b) {
    bool ia = affine_is_inf(a), ib = affine_is_inf(b);
    if (ia || ib) return ia && ib;
    return fp_eq(a.x, b.x) && fp_eq(a.y, b.y);
}

BP_HD affine affine_load(const affine_mem& m) {
    affine r;
    r.x = fp_from_words(m.x);
    r.y = fp_from_words(m.y);
    return r;
}
BP_HD affine_mem affine_store(const affine& p) {
    affine_mem m;
    if (affine_is_inf(p)) {
#pragma unroll
        for (int i = 0; i < 8; i++) m.x.w[i] = m.y.w[i] = 0;
        return m;
    }
    m.x = fp_to_words(p.x);
    m.y = fp_to_words(p.y);
    return m;
}
// 64-byte gather as four 16-byte loads.  Every array of affine_mem in this library starts on a 256-byte boundary (hipMalloc,
// the carved workspaces), so an element is 64-byte aligned; the struct itself only promises 4, which makes the compiler split
// a plain copy into overlapping dwordx3/x4 loads and patch the pieces together with moves that wait for the first piece right
// after it was issued.  Unconditional by design: software-pipelined loops call it outside any branch, so that the number of
// loads in flight is the same on every path and the waits are exact s_waitcnt vmcnt(N).
BP_HD affine_mem affine_mem_ld(const affine_mem* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4* q = reinterpret_cast<const uint4*>(__builtin_assume_aligned(p, 64));
    const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    affine_mem m;
    m.x.w[0] = a.x, m.x.w[1] = a.y, m.x.w[2] = a.z, m.x.w[3] = a.w;
    m.x.w[4] = b.x, m.x.w[5] = b.y, m.x.w[6] = b.z, m.x.w[7] = b.w;
    m.y.w[0] = c.x, m.y.w[1] = c.y, m.y.w[2] = c.z, m.y.w[3] = c.w;
    m.y.w[4] = d.x, m.y.w[5] = d.y, m.y.w[6] = d.z, m.y.w[7] = d.w;
    return m;
#else
    return *p;
#endif
}
BP_HD bool affine_mem_is_inf(const affine_mem& m) { return fp_words_is_zero(m.x) && fp_words_is_zero(m.y); }

BP_HD bool xyzz_is_inf(const xyzz& p) { return fp_raw_is_zero(p.ZZ); }
BP_HD xyzz xyzz_inf() {
    xyzz r;
    r.X = fp_zero();
    r.Y = fp_one();
    r.ZZ = fp_zero();
    r.ZZZ = fp_zero();
    return r;
}
BP_HD xyzz xyzz_from_affine(const affine& p) {
    xyzz r;
    if (affine_is_inf(p)) return xyzz_inf();
    r.X = p.x;
    r.Y = p.y;
    r.ZZ = fp_one();
    r.ZZZ = fp_one();
    return r;
}
BP_HD xyzz xyzz_neg(const xyzz& p) {
    xyzz r = p;
    r.Y = fp_neg(p.Y);
    return r;
}

// dbl-2008-s-1 with a = 0: 6M + 3S.  (secp256k1 has prime order: no point has Y = 0.)
BP_HD xyzz xyzz_dbl(const xyzz& p) {
    if (xyzz_is_inf(p)) return xyzz_inf();
    xyzz r;
    fp U = fp_dbl(p.Y);
    fp V = fp_sqr(U);
    fp W = fp_mul(U, V);
    fp S = fp_mul(p.X, V);
    fp XX = fp_sqr(p.X);
    fp M = fp_add(fp_dbl(XX), XX);
    r.X = fp_sub(fp_sqr(M), fp_dbl(S));
    r.Y = fp_sub(fp_mul(M, fp_sub(S, r.X)), fp_mul(W, p.Y));
    r.ZZ = fp_mul(V, p.ZZ);
    r.ZZZ = fp_mul(W, p.ZZZ);
    return r;
}

// doubling of an affine point (mdbl-2008-s-1; ZZ = ZZZ = 1).
BP_HD xyzz xyzz_dbl_affine(const affine& p) {
    if (affine_is_inf(p)) return xyzz_inf();
    xyzz r;
    fp U = fp_dbl(p.y);
    fp V = fp_sqr(U);
    fp W = fp_mul(U, V);
    fp S = fp_mul(p.x, V);
    fp XX = fp_sqr(p.x);
    fp M = fp_add(fp_dbl(XX), XX);
    r.X = fp_sub(fp_sqr(M), fp_dbl(S));
    r.Y = fp_sub(fp_mul(M, fp_sub(S, r.X)), fp_mul(W, p.y));
    r.ZZ = V;
    r.ZZZ = W;
    return r;
}

// acc += (neg ? -p : p) (p affine): madd-2008-s, 8M + 2S.  The sign goes into the multiplication by ZZZ as a lazy operand
// (nine subtractions and selects instead of a normalised negation behind a branch), X3 = R^2 - PPP - 2Q is one
// normalisation instead of three and Q - X3 none (fp.h, "linear combinations"): four normalisations per addition, not seven.
BP_HD void xyzz_madd_s(xyzz& acc, const affine& p, bool neg) {
    if (affine_is_inf(p)) return;
    if (xyzz_is_inf(acc)) {
        acc = xyzz_from_affine(p);
        if (neg) acc.Y = fp_neg(acc.Y);
        return;
    }
    fp U2 = fp_mul(p.x, acc.ZZ);
    fp S2 = fp_mul(fp_cneg_lazy(p.y, neg), acc.ZZZ);
    fp Pd = fp_sub(U2, acc.X);
    fp R = fp_sub(S2, acc.Y);
    if (fp_is_zero(Pd)) {
        if (fp_is_zero(R)) {
            affine q = p;
            if (neg) q.y = fp_neg(p.y);
            acc = xyzz_dbl_affine(q);
        } else {
            acc = xyzz_inf();
        }
        return;
    }
    fp PP = fp_sqr(Pd);
    fp PPP = fp_mul(Pd, PP);
    fp Qv = fp_mul(acc.X, PP);
    fp X3 = fp_sub_b_2c(fp_sqr(R), PPP, Qv);
    fp Y3 = fp_sub(fp_mul(R, fp_sub_lazy(Qv, X3)), fp_mul(acc.Y, PPP));
    acc.X = X3;
    acc.Y = Y3;
    acc.ZZ = fp_mul(acc.ZZ, PP);
    acc.ZZZ = fp_mul(acc.ZZZ, PPP);
}
BP_HD void xyzz_madd(xyzz& acc, const affine& p) { xyzz_madd_s(acc, p, false); }

// acc += q (both XYZZ): add-2008-s, 12M + 2S.
BP_HD void xyzz_add(xyzz& acc, const xyzz& q) {
    if (xyzz_is_inf(q)) return;
    if (xyzz_is_inf(acc)) {
        acc = q;
        return;
    }
    fp U1 = fp_mul(acc.X, q.ZZ);
    fp U2 = fp_mul(q.X, acc.ZZ);
    fp S1 = fp_mul(acc.Y, q.ZZZ);
    fp S2 = fp_mul(q.Y, acc.ZZZ);
    fp Pd = fp_sub(U2, U1);
    fp R = fp_sub(S2, S1);
    if (fp_is_zero(Pd)) {
        if (fp_is_zero(R))
            acc = xyzz_dbl(acc);
        else
            acc = xyzz_inf();
        return;
    }
    fp PP = fp_sqr(Pd);
    fp PPP = fp_mul(Pd, PP);
    fp Qv = fp_mul(U1, PP);
    fp X3 = fp_sub_b_2c(fp_sqr(R), PPP, Qv);
    fp Y3 = fp_sub(fp_mul(R, fp_sub_lazy(Qv, X3)), fp_mul(S1, PPP));
    acc.X = X3;
    acc.Y = Y3;
    acc.ZZ = fp_mul(fp_mul(acc.ZZ, q.ZZ), PP);
    acc.ZZZ = fp_mul(fp_mul(acc.ZZZ, q.ZZZ), PPP);
}

// 2^k * p through Jacobian coordinates: the serial doubling chains (Horner over the MSM windows) are
// latency-bound in one lane, and dbl-2009-l with a = 0 is 2M + 5S against 6M + 3S in XYZZ.
// XYZZ -> Jacobian with Z = ZZ: (X*ZZ, Y*ZZZ, ZZ)  [x = X*ZZ/ZZ^2, y = Y*ZZZ/ZZ^3, ZZ^3 = ZZZ^2];
// back: ZZ' = Z^2, ZZZ' = Z^3.  Worth it from k = 3.
BP_HD xyzz xyzz_dbl_n(const xyzz& p, int k) {
    if (xyzz_is_inf(p) || k <= 0) return p;
    if (k < 3) {
        xyzz r = p;
        for (int i = 0; i < k; i++) r = xyzz_dbl(r);
        return r;
    }
    fp X = fp_mul(p.X, p.ZZ), Y = fp_mul(p.Y, p.ZZZ), Z = p.ZZ;
    // dbl-2009-l with the linear steps merged (fp.h, "linear combinations"): six normalisations per doubling instead of
    // fourteen.  H = (X + B)^2 - A - C = D / 2.
    for (int i = 0; i < k; i++) {
        fp A = fp_sqr(X);
        fp B = fp_sqr(Y);
        fp C = fp_sqr(B);
        fp H = fp_sub2(fp_sqr(fp_add(X, B)), A, C);
        fp E = fp_mul3(A);
        fp Z3 = fp_mul(fp_dbl_lazy(Y), Z);
        X = fp_sub_4b(fp_sqr(E), H);                                              // F - 2D
        Y = fp_sub_2b(fp_mul(E, fp_2a_sub_lazy(H, X)), fp_mul4(C));               // E (D - X3) - 8C
        Z = Z3;
    }
    xyzz r;
    r.X = X;
    r.Y = Y;
    r.ZZ = fp_sqr(Z);
    r.ZZZ = fp_mul(r.ZZ, Z);
    return r;
}

// Affine form (one field inversion).  1/ZZZ = Z^-3, ZZ/ZZZ = Z^-1.
BP_HD affine xyzz_to_affine(const xyzz& p) {
    if (xyzz_is_inf(p)) return affine_inf();
    fp zi3 = fp_inv(p.ZZZ);
    fp zi = fp_mul(p.ZZ, zi3);
    fp zi2 = fp_sqr(zi);
    affine r;
    r.x = fp_mul(p.X, zi2);
    r.y = fp_mul(p.Y, zi3);
    return r;
}

// The same with the variable-time inversion (fp_inv_var): for the single-lane tail of an MSM, where the
// inversion's latency is exposed; throughput kernels keep xyzz_to_affine (uniform control flow).
BP_HD affine xyzz_to_affine_var(const xyzz& p) {
    if (xyzz_is_inf(p)) return affine_inf();
    fp zi3 = fp_inv_var(p.ZZZ);
    fp zi = fp_mul(p.ZZ, zi3);
    fp zi2 = fp_sqr(zi);
    affine r;
    r.x = fp_mul(p.X, zi2);
    r.y = fp_mul(p.Y, zi3);
    return r;
}

// y^2 == x^3 + 7 ?
BP_HD bool affine_on_curve(const affine& p) {
    fp rhs = fp_mul(fp_sqr(p.x), p.x);
    fp seven = fp_zero();
    seven.v[0] = 7;
    rhs = fp_add(rhs, seven);
    return fp_eq(fp_sqr(p.y), rhs);
}

// 33-byte SEC1 compressed -> affine (curv `Point::from_bytes`); false if not a curve point.
BP_HD bool affine_decompress(affine& out, const uint8_t* b33) {
    if (b33[0] != 2 && b33[0] != 3) return false;
    fp_words xw = fp_words_from_be(b33 + 1);
    if (!fp_words_is_canonical(xw)) return false;
    fp x = fp_from_words(xw);
    fp rhs = fp_mul(fp_sqr(x), x);
    fp seven = fp_zero();
    seven.v[0] = 7;
    rhs = fp_add(rhs, seven);
    fp y = fp_canon(fp_sqrt_candidate(rhs));
    if (!fp_eq(fp_sqr(y), rhs)) return false;
    if ((y.v[0] & 1u) != (uint32_t)(b33[0] & 1u)) y = fp_neg(y);
    out.x = x;
    out.y = y;
    return true;
}

// memory form with validation: canonical coordinates and on the curve (infinity allowed only if asked)
BP_HD bool affine_mem_valid(const affine_mem& m, bool allow_inf) {
    if (affine_mem_is_inf(m)) return allow_inf;
    if (!fp_words_is_canonical(m.x) || !fp_words_is_canonical(m.y)) return false;
    return affine_on_curve(affine_load(m));
}

// boundary encoding: x||y big-endian, 64 zero bytes = infinity (include/bpmsm.h)
BP_HD void affine_to_be64(uint8_t* b, const affine& p) {
    affine_mem m = affine_store(p);
    fp_words_to_be(b, m.x);
    fp_words_to_be(b + 32, m.y);
}
BP_HD affine affine_from_be64(const uint8_t* b) {
    affine r;
    r.x = fp_from_be(b);
    r.y = fp_from_be(b + 32);
    return r;
}
BP_HD affine_mem affine_mem_from_be64(const uint8_t* b) {
    affine_mem m;
    m.x = fp_words_from_be(b);
    m.y = fp_words_from_be(b + 32);
    return m;
}

}  // namespace bp
