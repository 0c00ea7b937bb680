// Quad-cooperative point operations for the latency-bound tails of the MSM (bucket-reduction trees, window scans, the
// Horner chain; reference call site: the fold of /root/reference/src/proofs/range_proof.rs:770-774 at 2^18..2^22 terms).
//
// A lone wave pays ~2 ns per VALU instruction whatever the number of active lanes, so a serial XYZZ addition (14 field
// multiplications, ~3000 instructions) costs ~5.4 us and the trees / chains built from it are pure latency while 60 of the
// wave's 64 lanes idle.  Here the FOUR lanes of a quad hold the same point (replicated registers) and share one operation:
// the independent multiplications of a formula level go one to each lane (operand select by lane id: v_cndmask), the
// products come back to all four with DPP quad_perm broadcasts (no LDS, no barrier), the cheap linear steps are done
// redundantly.  An addition is 4 multiplication levels instead of 14 multiplications (~2.2x less latency), a Jacobian
// doubling 3 instead of 7.  All four lanes of a quad must be active and hold identical operands.
// Same group elements as ec.h's formulas (the affine result is canonical), so every byte that leaves is unchanged.
#pragma once
#include "ec.h"

#if defined(__HIPCC__)
namespace bp {

template <int Q>
__device__ __forceinline__ fp fpq_bcast(const fp& a) {
    fp r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
#if defined(__HIP_DEVICE_COMPILE__)
        r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[i], Q * 0x55, 0xf, 0xf, true);
#else
        r.v[i] = a.v[i];  // host pass of hipcc: declaration only
#endif
    }
    return r;
}
__device__ __forceinline__ fp fpq_sel(int q, const fp& a0, const fp& a1, const fp& a2, const fp& a3) {
    fp r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const uint32_t lo = (q & 1) ? a1.v[i] : a0.v[i];
        const uint32_t hi = (q & 1) ? a3.v[i] : a2.v[i];
        r.v[i] = (q & 2) ? hi : lo;
    }
    return r;
}
// r_j = a_j * b_j for j = 0..3, lane j of the quad computing product j; at most one operand of a pair may be lazy (fp.h)
__device__ __forceinline__ void fpq_mul4(int q, const fp& a0, const fp& b0, const fp& a1, const fp& b1, const fp& a2, const fp& b2, const fp& a3,
                                         const fp& b3, fp& r0, fp& r1, fp& r2, fp& r3) {
    const fp p = fp_mul(fpq_sel(q, a0, a1, a2, a3), fpq_sel(q, b0, b1, b2, b3));
    r0 = fpq_bcast<0>(p);
    r1 = fpq_bcast<1>(p);
    r2 = fpq_bcast<2>(p);
    r3 = fpq_bcast<3>(p);
}
__device__ __forceinline__ void fpq_mul3(int q, const fp& a0, const fp& b0, const fp& a1, const fp& b1, const fp& a2, const fp& b2, fp& r0, fp& r1,
                                         fp& r2) {
    const fp p = fp_mul(fpq_sel(q, a0, a1, a2, a2), fpq_sel(q, b0, b1, b2, b2));
    r0 = fpq_bcast<0>(p);
    r1 = fpq_bcast<1>(p);
    r2 = fpq_bcast<2>(p);
}

// acc += o (both XYZZ, replicated over the quad): add-2008-s as in ec.h's xyzz_add, four multiplication levels
__device__ __forceinline__ void xyzzq_add(int q, xyzz& acc, const xyzz& o) {
    if (xyzz_is_inf(o)) return;  // uniform over the quad: the operands are replicated
    if (xyzz_is_inf(acc)) {
        acc = o;
        return;
    }
    fp U1, U2, S1, S2;
    fpq_mul4(q, acc.X, o.ZZ, o.X, acc.ZZ, acc.Y, o.ZZZ, o.Y, acc.ZZZ, U1, U2, S1, S2);
    const fp Pd = fp_sub(U2, U1);
    const fp R = fp_sub(S2, S1);
    if (fp_is_zero(Pd)) {
        if (fp_is_zero(R))
            acc = xyzz_dbl(acc);  // rare: every lane of the quad doubles its own copy
        else
            acc = xyzz_inf();
        return;
    }
    fp PP, RR, Z2, Z3;
    fpq_mul4(q, Pd, Pd, R, R, acc.ZZ, o.ZZ, acc.ZZZ, o.ZZZ, PP, RR, Z2, Z3);
    fp PPP, Qv, ZZn;
    fpq_mul3(q, Pd, PP, U1, PP, Z2, PP, PPP, Qv, ZZn);
    const fp X3 = fp_sub_b_2c(RR, PPP, Qv);
    fp A, B, ZZZn;
    fpq_mul3(q, R, fp_sub_lazy(Qv, X3), S1, PPP, Z3, PPP, A, B, ZZZn);
    acc.X = X3;
    acc.Y = fp_sub(A, B);
    acc.ZZ = ZZn;
    acc.ZZZ = ZZZn;
}

// 2^k * p through Jacobian coordinates as ec.h's xyzz_dbl_n (dbl-2009-l, a = 0): per doubling two shared levels of three
// multiplications and one multiplication every lane does for itself
__device__ __forceinline__ xyzz xyzzq_dbl_n(int q, const xyzz& p, int k) {
    if (xyzz_is_inf(p) || k <= 0) return p;
    fp X, Y, dummy;
    fpq_mul3(q, p.X, p.ZZ, p.Y, p.ZZZ, p.ZZ, p.ZZ, X, Y, dummy);
    fp Z = p.ZZ;
    for (int i = 0; i < k; i++) {
        fp A, B, Z3;
        fpq_mul3(q, X, X, Y, Y, fp_dbl_lazy(Y), Z, A, B, Z3);
        const fp E = fp_mul3(A);
        fp C, T, F;
        fpq_mul3(q, B, B, fp_add(X, B), fp_add(X, B), E, E, C, T, F);
        const fp H = fp_sub2(T, A, C);
        X = fp_sub_4b(F, H);                                             // F - 2D, D = 2H
        Y = fp_sub_2b(fp_mul(E, fp_2a_sub_lazy(H, X)), fp_mul4(C));      // E (D - X3) - 8C
        Z = Z3;
    }
    xyzz r;
    r.X = X;
    r.Y = Y;
    r.ZZ = fp_sqr(Z);
    r.ZZZ = fp_mul(r.ZZ, Z);
    return r;
}

}  // namespace bp
#endif
