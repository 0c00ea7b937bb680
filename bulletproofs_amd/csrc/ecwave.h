// Lane trees over XYZZ partial sums without LDS round trips (north_star: "wavefront-shuffle bucket reduction"; reference call
// site: the fold of /root/reference/src/proofs/range_proof.rs:770-774, whose per-lane partial sums end in one point).
//
// The G lanes of a group (G a power of two, the group aligned to G lanes) each hold a partial sum; lane `sub` of the group adds
// the partial of lane sub + off for off = G/2 .. 1.  Inside a wave the 36 words of the partner's point come through the
// cross-lane network - DPP row shifts (inside a row of 16 lanes, VALU rate) for the one- and two-level trees of the large-batch
// kernels, ds_bpermute (LDS crossbar, no LDS memory) for the longer ones - so a level is 36 register moves and one addition: no
// store, no barrier, no load.  Only the levels
// that cross waves (off >= 64, groups of 128 or 256 lanes) still go through LDS.  Every lane of the wave must reach the call
// (the moves read the partner's registers); the sum is valid in the lane with sub == 0.
#pragma once
#include "ec.h"
#include "ecq.h"

#if defined(__HIPCC__)
namespace bp {

// value of `x` in lane (own lane + OFF); unspecified (never used by the trees) where that lane lies outside the row / wave
template <int OFF>
__device__ __forceinline__ uint32_t wave_down32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (OFF < 16) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x100 + OFF, 0xf, 0xf, true);  // row_shl:OFF
    } else {
        const int src = (int)((threadIdx.x & 63) + OFF) & 63;
        return (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)x);
    }
#else
    return x;  // host pass of hipcc: declaration only
#endif
}
template <int OFF>
__device__ __forceinline__ fp fp_wave_down(const fp& a) {
    fp r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = wave_down32<OFF>(a.v[i]);
    return r;
}
template <int OFF>
__device__ __forceinline__ void xyzz_wave_level(xyzz& acc, int sub) {
    xyzz o;
    o.X = fp_wave_down<OFF>(acc.X);
    o.Y = fp_wave_down<OFF>(acc.Y);
    o.ZZ = fp_wave_down<OFF>(acc.ZZ);
    o.ZZZ = fp_wave_down<OFF>(acc.ZZZ);
    if (sub < OFF) xyzz_add(acc, o);
}
// the same with a run-time distance (any 1..32): ds_bpermute for every level, ONE addition in the code however many levels loop
// through it (unrolled DPP levels cost a copy of the ~3000-instruction addition each: k_bpv_fix<32> went over its register budget
// and the library grew by half when all trees were unrolled)
__device__ __forceinline__ void xyzz_wave_level_rt(xyzz& acc, int sub, int off) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int src = ((int)((threadIdx.x & 63) + off) & 63) << 2;
    xyzz o;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        o.X.v[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)acc.X.v[i]);
        o.Y.v[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)acc.Y.v[i]);
        o.ZZ.v[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)acc.ZZ.v[i]);
        o.ZZZ.v[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)acc.ZZZ.v[i]);
    }
    if (sub < off) xyzz_add(acc, o);
#endif
}
// levels G/2 .. 1 of a group that lies inside one wave (G <= 64): one or two levels unrolled on DPP moves, longer trees as a loop
template <int G>
__device__ __forceinline__ void xyzz_wave_tree(xyzz& acc, int sub) {
    static_assert(G >= 1 && G <= 64 && (G & (G - 1)) == 0, "group inside a wave");
    if constexpr (G == 4) xyzz_wave_level<2>(acc, sub);
    if constexpr (G == 2 || G == 4) xyzz_wave_level<1>(acc, sub);
    if constexpr (G > 4) {
#pragma nounroll
        for (int off = G / 2; off >= 1; off >>= 1) xyzz_wave_level_rt(acc, sub, off);
    }
}
// groups of up to 256 lanes of a 256-thread workgroup: the levels that cross waves through `red` (256 entries, only touched when
// G > 64; every thread of the workgroup must call), the rest in the wave
template <int G>
__device__ __forceinline__ void xyzz_group_tree(xyzz& acc, int sub, int tid, xyzz* red) {
    if constexpr (G > 64) {
        red[tid] = acc;
        __syncthreads();
#pragma unroll
        for (int off = G / 2; off >= 128; off >>= 1) {
            if (sub < off) {
                xyzz v = red[tid];
                xyzz o = red[tid + off];
                xyzz_add(v, o);
                red[tid] = v;
            }
            __syncthreads();
        }
        if (sub < 64) {  // uniform over a wave: groups are whole waves here
            acc = red[tid];
            xyzz o = red[tid + 64];
            xyzz_add(acc, o);
            xyzz_wave_tree<64>(acc, sub);
        }
    } else {
        xyzz_wave_tree<G>(acc, sub);
    }
}

// Latency-bound trees (one workgroup alone on its CU: the single-proof MSMs, the small-batch verifier): from the level at which
// a group has four lanes per addition left, the additions are shared by QUADS (ecq.h: 2.2 times less latency than a lane's own
// addition).  Group of G lanes (G = 64 .. 256, a divisor of the 256-thread workgroup, groups aligned to G), lane `sub` of its
// group, `gbase` = the group's first slot in `red` (256 entries).  Level 1: the lower half adds the upper half, one addition
// per lane; then G/4 quads halve the G/2 partials level by level, operands replicated over a quad by LDS broadcast reads.
// Every thread of the workgroup must call; the sum is valid in the lanes of the group's first quad (sub < 4).
template <int G>
__device__ __forceinline__ void xyzz_group_tree_quad(xyzz& acc, int sub, int gbase, xyzz* red) {
    static_assert(G >= 8 && G <= 256 && (G & (G - 1)) == 0, "group of 8..256 lanes");
    red[gbase + sub] = acc;
    __syncthreads();
    if (sub < G / 2) {
        xyzz v = red[gbase + sub];
        xyzz o = red[gbase + sub + G / 2];
        xyzz_add(v, o);
        red[gbase + sub] = v;
    }
    __syncthreads();
    const int q = sub & 3, j = sub >> 2;
#pragma nounroll
    for (int cnt = G / 4; cnt >= 1; cnt >>= 1) {
        if (j < cnt) {
            acc = red[gbase + j];
            xyzz o = red[gbase + j + cnt];
            xyzzq_add(q, acc, o);
            if (q == 0 && cnt > 1) red[gbase + j] = acc;
        }
        __syncthreads();
    }
}

}  // namespace bp
#endif
