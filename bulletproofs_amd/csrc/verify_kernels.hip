// Batched single-MSM verifier for Bulletproofs range proofs: a device-side restatement of
// `RangeProof::aggregated_verify` (/root/reference/src/proofs/range_proof.rs:537-782), one proof per
// group of lanes, thousands of proofs per launch.
//
//   k_bpv_decompress : one lane per proof point.  33-byte SEC1 -> affine limbs (curv
//                      `Point::from_bytes`), commitments ingested with the on-curve check.
//   (verify_scalars.hip): Fiat-Shamir challenges y, z, x, x_u, c, x_j (range_proof.rs:565-594,
//                      650-668) with SHA-256 on device, the two fixed-base multiplications y*G and
//                      x_u*G (:571, :588) from the generator table, then the whole scalar vector of
//                      :670-740 in Montgomery arithmetic mod q, recoded to signed window digits.
//   k_bpv_fix        : the 2nm+2 fixed-generator terms of the MSM of :742-774: table gathers + mixed additions, no
//                      doublings; one partial sum per proof.
//   k_bpv_msm        : TP lanes per proof.  The proof-specific points (S, L_j, R_j, V_j, T1, T2) via GLV +
//                      interleaved 4-bit windows (ecmul.h: lanes that own several points share one
//                      doubling chain), lane-tree reduction, + the fixed partial, and the comparison with A (:777).
//   k_bpv_small      : latency-bound batches (up to ~1024 proofs): both halves in one launch, four lanes per
//                      proof-specific point (k_bpv_msm_slice is its two-launch form for shapes beyond 128 such lanes).
//
// Per-proof verdicts follow `Result<(), Errors>`: 0 = Ok, 3 = RangeProofError, 255 = the proof bytes
// would not deserialize in the reference (bad point / scalar >= q / zero challenge).
#include <hip/hip_runtime.h>

#include "ec.h"
#include "ecmul.h"
#include "ecwave.h"
#include "fq.h"
#include "kernels.h"

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

// Bulletproofs+ proofs (A | L[k] | R[k] | a_tag b_tag | r' s' delta', oracle/encoding.py) into the point order
// b_tag, A, L[k], R[k], a_tag, V[m]: the MSM's variable terms are then pts[1..] and the comparand is pts[0].
__global__ void k_wipv_decompress(const uint8_t* __restrict__ proofs, size_t proof_len, const uint8_t* __restrict__ coms_xy, int k, int m,
                                  size_t nproofs, affine_mem* __restrict__ pts, int* __restrict__ bad) {
    const int npc = 3 + 2 * k;
    const int NPT = npc + m;
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nproofs * (size_t)NPT) return;
    size_t pr = gid / NPT;
    int j = (int)(gid % NPT);  // index in pts
    affine a;
    bool ok;
    if (j < npc) {
        const int f = (j == 0) ? 2 * k + 2 : j - 1;  // position in the proof: A, L.., R.., a_tag, b_tag
        const uint8_t* src = proofs + pr * proof_len + (size_t)33 * f;
        uint8_t b[33];
        for (int i = 0; i < 33; i++) b[i] = src[i];
        ok = affine_decompress(a, b);
        if (!ok) a = affine_inf();
    } else {
        const uint8_t* src = coms_xy + (pr * (size_t)m + (j - npc)) * 64;
        uint8_t b[64];
        for (int i = 0; i < 64; i++) b[i] = src[i];
        affine_mem mm = affine_mem_from_be64(b);
        ok = affine_mem_valid(mm, false);
        a = ok ? affine_load(mm) : affine_inf();
    }
    pts[pr * NPT + j] = affine_store(a);
    if (!ok) atomicOr(&bad[pr], 1);
}

// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Fixed-generator half of the verification MSM (the 2nm+2 terms over g_vec, h_vec, G, H; range_proof.rs:742-774): table
// gathers + mixed additions only, no doublings.  A kernel of its own because the loop needs ~120 VGPRs, so it runs at 4
// waves per SIMD where the proof-specific half (233 VGPRs) runs at 2, and because the next table entry can be fetched
// while the current one is being added: the 64-byte gather from a table of tens of GiB is an HBM round trip, hidden
// behind a 3.4 us mixed addition instead of exposed in front of it.  TP lanes per proof, LDS tree, one XYZZ partial
// per proof in `fixpart`, which k_bpv_msm adds to its own sum.
#ifndef BPV_FIX_WAVES
#define BPV_FIX_WAVES 3
#endif
// one lane's share of a proof's fixed-generator terms (terms lane, lane + TP, ...): table gathers + mixed additions
template <int TP>
__device__ __forceinline__ xyzz bpv_fix_lane_sum(int lane, int nm, const affine_mem* __restrict__ table, int c, int W, const tdig_t* __restrict__ dg,
                                                 int shiftW) {
    const int NB = 1 << (c - 1);
    xyzz acc = xyzz_inf();
    const int nfw = (2 * nm + 2) * W;
    // shiftW != 0: the proof uses the first nm of the table's nm_tab generators per vector (a batch mixing aggregation
    // sizes over one bp_gens): h_vec's rows start (nm_tab - nm) rows later, G and H twice that
    const int hW = nm * W, gW = 2 * nm * W;
    // Software pipeline, two terms per trip through two entry buffers (A, B): while term idx is added, the table entry of
    // term idx + TP and the digit of term idx + 2 TP are in flight.  Every load is UNCONDITIONAL (a term past the end or a
    // zero digit reads a valid dummy address and is masked at the addition): with loads inside `if (d != 0)` the number of
    // loads in flight differed per path and the compiler fell back to s_waitcnt vmcnt(0) in front of every addition, which
    // waited for the prefetch it had just issued (ISA of round 2's first version).  Measured on the 65 536-proof batch: no
    // change (7.22 -> 7.24 ms) - at three waves per SIMD the other waves already covered the exposed gather; it is what a
    // lone wave of a small batch needs.
    auto digit = [&](int idx) { return (int)dg[idx < nfw ? idx : nfw - 1]; };
    auto entry = [&](int idx, int d) {
        const int i = idx < nfw ? idx : nfw - 1;
        const int row = i + (i >= hW ? shiftW : 0) + (i >= gW ? shiftW : 0);
        const int a = d < 0 ? -d : d;
        return affine_mem_ld(table + (size_t)row * NB + (a ? a - 1 : 0));
    };
    auto add = [&](const affine_mem& m, int d, int idx) {
        if (d != 0 && idx < nfw) {
            affine e = affine_load(m);
            xyzz_madd_s(acc, e, d < 0);
        }
    };
    int dA = digit(lane), dB = digit(lane + TP);
    affine_mem A = entry(lane, dA), B;
    for (int idx = lane; idx < nfw; idx += 2 * TP) {
        int dN = digit(idx + 2 * TP);
        B = entry(idx + TP, dB);
        add(A, dA, idx);
        dA = dN;
        dN = digit(idx + 3 * TP);
        A = entry(idx + 2 * TP, dA);
        add(B, dB, idx + TP);
        dB = dN;
    }
    return acc;
}

template <int TP>
__global__ __launch_bounds__(256, BPV_FIX_WAVES) void k_bpv_fix(int nm, size_t nproofs, const affine_mem* __restrict__ table, int c, int W,
                                                    const tdig_t* __restrict__ digF, const int* __restrict__ bad, int shiftW,
                                                    xyzz* __restrict__ fixpart) {
    // Trees of one or two levels (the large batches) on DPP moves (ecwave.h); the longer ones stay in LDS (ds_bpermute levels
    // measured no gain here: the tree is 5 of this kernel's ~75 additions per lane and three waves per SIMD hide the barriers)
    __shared__ xyzz red[TP > 4 ? 256 : 1];
    const int tid = threadIdx.x;
    const int lane = tid % TP;
    const size_t pr = (size_t)blockIdx.x * (256 / TP) + tid / TP;
    const bool active = pr < nproofs && !bad[pr];
    xyzz acc = xyzz_inf();
    if (active) acc = bpv_fix_lane_sum<TP>(lane, nm, table, c, W, digF + pr * (size_t)((2 * nm + 2) * W), shiftW);
    if constexpr (TP > 4) {
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
        acc = red[tid];
    } else {
        xyzz_wave_tree<TP>(acc, lane);
    }
    if (lane == 0 && pr < nproofs) fixpart[pr] = acc;
}

#ifndef BPV_MSM_WAVES
#define BPV_MSM_WAVES 2
#endif

// TP lanes per proof, PPL proof-specific points per lane and pass (one shared doubling chain per
// lane: the fewer lanes a proof has, the fewer chains it pays for); 256-thread workgroups.
template <int TP, int PPL>
__global__ __launch_bounds__(256, BPV_MSM_WAVES) void k_bpv_msm(int n_bits, int m, int k, size_t nproofs, const affine_mem* __restrict__ pts,
                                                 const xyzz* __restrict__ fixpart,
                                                 const fq* __restrict__ varS, xyzz* __restrict__ vartab, affine_mem* __restrict__ varaff,
                                                 int8_t* __restrict__ vardig, const int* __restrict__ bad, uint8_t* __restrict__ verdicts,
                                                 int wip, xyzz* __restrict__ varpart) {
    // varpart != nullptr: k_bpv_fix runs BESIDE this kernel on the helper stream (launch_bpv_msm), so the sum does not start from
    // fixpart[pr]; the proof-specific sum alone goes to varpart[pr] and k_bpv_verdict finishes
    __shared__ xyzz red[TP > 64 ? 256 : 1];
    const int tid = threadIdx.x;
    const int lane = tid % TP;
    const size_t pr = (size_t)blockIdx.x * (256 / TP) + tid / TP;
    const bool active = pr < nproofs && !(pr < nproofs && bad[pr]);
    // Bulletproofs: pts = A S T1 T2 L[k] R[k] V[m], 3+2k+m variable terms, compared with A.
    // Bulletproofs+ (wip): pts = b_tag A L[k] R[k] a_tag V[m], variable terms pts[1..], compared with b_tag.
    const int NPT = (wip ? 3 : 4) + 2 * k + m;
    const int NV = (wip ? 2 : 3) + 2 * k + m;
    xyzz acc = xyzz_inf();
    if (active) {
        if (lane == 0 && !varpart) acc = fixpart[pr];  // the fixed-generator half, summed by k_bpv_fix
        const affine_mem* P = pts + pr * NPT;
        const fq* vs = varS + pr * (size_t)NV;
        const size_t gl = (size_t)blockIdx.x * 256 + tid;
        xyzz* tab = vartab + gl * (PPL * ECMUL_SLOTS);
        affine_mem* atab = varaff + gl * (PPL * ECMUL_NT);
        int8_t* dig = vardig + gl * (PPL * ECMUL_DIGB);
        for (int v0 = lane; v0 < NV; v0 += TP * PPL) {
            int cnt = (NV - v0 + TP - 1) / TP;
            if (cnt > PPL) cnt = PPL;
            // variable point order: S, L_0.., R_0.., V_0.., T1, T2  ->  index into pts
            auto loadP = [&](int j) {
                int v = v0 + TP * j;
                int pi = wip ? 1 + v : (v == 0) ? 1 : (v < 1 + 2 * k + m) ? (3 + v) : (v == 1 + 2 * k + m ? 2 : 3);
                return affine_load(P[pi]);
            };
            auto loadK = [&](int j) { return vs[v0 + TP * j]; };
            xyzz t = ec_mul_glv_multi<PPL>(loadP, loadK, cnt, tab, atab, dig);
            xyzz_add(acc, t);
        }
    }
    xyzz_group_tree<TP>(acc, lane, tid, red);
    if (lane == 0 && pr < nproofs && varpart) {
        varpart[pr] = acc;
    } else if (lane == 0 && pr < nproofs) {
        uint8_t verdict;
        if (bad[pr]) {
            verdict = 255;
        } else {
            // lhs == A ?  (range_proof.rs:777; == b_tag, range_proof_wip.rs:604) compared projectively: X = A.x*ZZ, Y = A.y*ZZZ
            xyzz r = acc;
            affine A = affine_load(pts[pr * NPT]);
            bool eq = !xyzz_is_inf(r) && fp_eq(r.X, fp_mul(A.x, r.ZZ)) && fp_eq(r.Y, fp_mul(A.y, r.ZZZ));
            verdict = eq ? 0 : 3;
        }
        verdicts[pr] = verdict;
    }
}

// Small batches (latency-bound: too few proofs to fill the SIMDs with one ladder per lane): FOUR lanes per proof-specific point
// (ec_mul_glv_slice: GLV half x window range, the upper range on the base 2^68 P), LPP lanes per proof.  The longest lane does
// 128 doublings + 17 additions instead of 128 doublings + 66 additions, and a 512-proof batch fills 512 waves instead of 128.
// More total work than k_bpv_msm (every slice pays for its own doubling chain), so only chosen when the batch cannot fill
// two waves per SIMD anyway (bpv_slice_lanes).
template <int LPP>
__global__ __launch_bounds__(256) void k_bpv_msm_slice(int m, int k, size_t nproofs, const affine_mem* __restrict__ pts,
                                                        const xyzz* __restrict__ fixpart, const fq* __restrict__ varS, xyzz* __restrict__ vartab,
                                                        const int* __restrict__ bad, uint8_t* __restrict__ verdicts, int wip,
                                                        xyzz* __restrict__ varpart) {  // varpart: as in k_bpv_msm
    __shared__ xyzz red[LPP > 64 ? 256 : 1];
    const int tid = threadIdx.x;
    const int lane = tid % LPP;
    const size_t pr = (size_t)blockIdx.x * (256 / LPP) + tid / LPP;
    const bool active = pr < nproofs && !bad[pr];
    const int NPT = (wip ? 3 : 4) + 2 * k + m;
    const int NV = (wip ? 2 : 3) + 2 * k + m;
    xyzz acc = xyzz_inf();
    if (active) {
        if (lane == 0 && !varpart) acc = fixpart[pr];
        const affine_mem* P = pts + pr * NPT;
        const fq* vs = varS + pr * (size_t)NV;
        xyzz* tab = vartab + ((size_t)blockIdx.x * 256 + tid) * ECMUL_NT;
        for (int idx = lane; idx < 4 * NV; idx += LPP) {
            const int v = idx >> 2, slice = idx & 3;
            // variable point order: S, L_0.., R_0.., V_0.., T1, T2  ->  index into pts (as in k_bpv_msm)
            const int pi = wip ? 1 + v : (v == 0) ? 1 : (v < 1 + 2 * k + m) ? (3 + v) : (v == 1 + 2 * k + m ? 2 : 3);
            xyzz t = ec_mul_glv_slice(affine_load(P[pi]), vs[v], slice, tab);
            xyzz_add(acc, t);
        }
    }
    xyzz_group_tree<LPP>(acc, lane, tid, red);
    if (lane == 0 && pr < nproofs && varpart) {
        varpart[pr] = acc;
    } else if (lane == 0 && pr < nproofs) {
        uint8_t verdict;
        if (bad[pr]) {
            verdict = 255;
        } else {
            xyzz r = acc;
            affine A = affine_load(pts[pr * NPT]);
            bool eq = !xyzz_is_inf(r) && fp_eq(r.X, fp_mul(A.x, r.ZZ)) && fp_eq(r.Y, fp_mul(A.y, r.ZZZ));
            verdict = eq ? 0 : 3;
        }
        verdicts[pr] = verdict;
    }
}

// verdict of a proof from the two partial sums when k_bpv_fix ran beside the proof-specific kernel: fixpart + varpart == A ?
// (range_proof.rs:777; == b_tag, range_proof_wip.rs:604)
__global__ __launch_bounds__(256) void k_bpv_verdict(int m, int k, size_t nproofs, const affine_mem* __restrict__ pts, const xyzz* __restrict__ fixpart,
                                                     const xyzz* __restrict__ varpart, const int* __restrict__ bad, uint8_t* __restrict__ verdicts,
                                                     int wip) {
    const size_t pr = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (pr >= nproofs) return;
    if (bad[pr]) {
        verdicts[pr] = 255;
        return;
    }
    const int NPT = (wip ? 3 : 4) + 2 * k + m;
    xyzz r = fixpart[pr];
    xyzz v = varpart[pr];
    xyzz_add(r, v);
    affine A = affine_load(pts[pr * NPT]);
    bool eq = !xyzz_is_inf(r) && fp_eq(r.X, fp_mul(A.x, r.ZZ)) && fp_eq(r.Y, fp_mul(A.y, r.ZZZ));
    verdicts[pr] = eq ? 0 : 3;
}

// Small batches, both halves of the MSM in ONE launch: 2 * LPP lanes per proof - the first LPP run the slices of
// k_bpv_msm_slice, the other LPP the fixed-generator terms of k_bpv_fix<LPP> - and one tree over all of them.  In a
// latency-bound batch the two kernels ran one after the other on a mostly empty GPU (512 proofs: 197 us of table
// additions, then 590 us of ladders); side by side the table additions disappear behind the ladders.  For LPP >= 64 the
// two roles are whole waves; below that a wave runs both paths in turn (no gain, no loss against two launches).
template <int LPP>
__global__ __launch_bounds__(256) void k_bpv_small(int nm, int m, int k, size_t nproofs, const affine_mem* __restrict__ pts,
                                                    const affine_mem* __restrict__ table, int c, int W, const tdig_t* __restrict__ digF, int shiftW,
                                                    const fq* __restrict__ varS, xyzz* __restrict__ vartab, const int* __restrict__ bad,
                                                    uint8_t* __restrict__ verdicts, int wip, int pre) {
    static_assert(2 * LPP <= 256, "one proof fits a workgroup");
    __shared__ xyzz red[256];
    const int tid = threadIdx.x;
    const int sub = tid % (2 * LPP), lane = sub % LPP;
    const bool fixrole = sub >= LPP;
    const size_t pr = (size_t)blockIdx.x * (256 / (2 * LPP)) + tid / (2 * LPP);
    const bool active = pr < nproofs && !bad[pr];
    const int NPT = (wip ? 3 : 4) + 2 * k + m;
    const int NV = (wip ? 2 : 3) + 2 * k + m;
    xyzz acc = xyzz_inf();
    if (active) {
        if (fixrole) {
            acc = bpv_fix_lane_sum<LPP>(lane, nm, table, c, W, digF + pr * (size_t)((2 * nm + 2) * W), shiftW);
        } else {
            const affine_mem* P = pts + pr * NPT;
            const fq* vs = varS + pr * (size_t)NV;
            xyzz* tab = vartab + (pr * LPP + lane) * ECMUL_NT;
            const int nparts = pre ? pre : 2, nsl = 2 * nparts;  // slices per point (pre: the window ranges the tables were built for)
            for (int idx = lane; idx < nsl * NV; idx += LPP) {
                const int v = idx / nsl, slice = idx % nsl;
                // variable point order: S, L_0.., R_0.., V_0.., T1, T2  ->  index into pts (as in k_bpv_msm)
                const int pi = wip ? 1 + v : (v == 0) ? 1 : (v < 1 + 2 * k + m) ? (3 + v) : (v == 1 + 2 * k + m ? 2 : 3);
                // pre: the slice's window table was built beside the constants kernel (k_bpv_consts_pre, verify_scalars.hip)
                const affine Pv = affine_load(P[pi]);
                xyzz t = !pre ? ec_mul_glv_slice(Pv, vs[v], slice, tab) : affine_is_inf(Pv) ? xyzz_inf() : ec_slice_ladder(vs[v], slice, tab, nparts);
                xyzz_add(acc, t);
            }
        }
    }
    xyzz_group_tree_quad<2 * LPP>(acc, sub, tid - sub, red);  // latency-bound: quad-shared additions from the second level on
    if (sub == 0 && pr < nproofs) {
        uint8_t verdict;
        if (bad[pr]) {
            verdict = 255;
        } else {
            xyzz r = acc;
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
hipError_t launch_wipv_decompress(const uint8_t* proofs, size_t proof_len, const uint8_t* coms_xy, int k, int m, size_t nproofs, affine_mem* pts,
                                  int* bad, hipStream_t st) {
    size_t total = nproofs * (size_t)(3 + 2 * k + m);
    hipLaunchKernelGGL(k_wipv_decompress, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, proofs, proof_len, coms_xy, k, m, nproofs, pts,
                       bad);
    return hipGetLastError();
}
// lanes per proof -> proof-specific points per lane and pass
static int bpv_ppl(int tp) { return tp == 2 ? 8 : (tp == 4 ? 4 : (tp == 8 ? 2 : 1)); }
// lanes per proof of the four-lanes-per-point kernel for NV proof-specific points (a power of two, at most one workgroup)
int bpv_slice_lanes(int nv, int nparts) {
    int lpp = 16;
    while (lpp < 2 * nparts * nv && lpp < 256) lpp *= 2;
    return lpp;
}
// tp > 0: lanes per proof of k_bpv_msm; tp < 0: k_bpv_msm_slice with -tp lanes per proof
size_t bpv_msm_vartab_bytes(int tp, size_t nproofs) {
    if (tp < 0) {
        size_t blocks = (nproofs * (size_t)(-tp) + 255) / 256;
        return blocks * 256 * ECMUL_NT * sizeof(xyzz);
    }
    size_t blocks = (nproofs + (256 / tp) - 1) / (256 / tp);
    return blocks * 256 * bpv_ppl(tp) * (ECMUL_SLOTS * sizeof(xyzz) + ECMUL_NT * sizeof(affine_mem) + ECMUL_DIGB);
}
size_t bpv_msm_fixpart_bytes(size_t nproofs) { return 2 * nproofs * sizeof(xyzz); }  // fixed-generator sums | proof-specific sums (helper-stream form)
// lanes per proof of the fixed-generator kernel: enough for 4 waves on every SIMD (262144 lanes) when the batch allows, 64 lanes
// only for batches of up to 2048 proofs (a 4096-proof batch spent more in the 64-lane tree than it won: 1.77 -> 1.54 ms per call in
// flight at 32 lanes; profiles/r02_small_batch.txt F)
static int bpv_fix_lanes(int tp, size_t nproofs, int nm) {
    static const int forced = bp_tune("BP_FIX_LANES") ? atoi(bp_tune("BP_FIX_LANES")) : 0;  // tuning sweeps
    if (forced == 2 || forced == 4 || forced == 8 || forced == 16 || forced == 32 || forced == 64) return forced;
    int tf = tp < 4 ? 4 : tp;
    while (tf < 64 && nproofs * (size_t)tf < 262144) tf *= 2;
    if (tf == 64 && nproofs > 2048 && tp < 64 && nm <= 128) tf = 32;  // (aggregated proofs have thousands of terms per proof: the tree is nothing)
    while (tf > 2 && tf > 2 * nm + 2) tf /= 2;
    return tf;
}
int bpv_small_fused(int tp) {
    static const bool split = bp_tune("BP_VERIFY_SMALL_SPLIT") != nullptr;  // A/B: the two-launch form
    return (tp < 0 && -tp <= 128 && !split) ? -tp : 0;
}
// window ranges per GLV half in the fused small-batch kernels: four when the lanes per proof cover eight slices per point
// (a power of two >= 4 nv is always < 8 nv, so the lane count alone tells), else two
int bpv_small_parts(int tp, int nv) {
    const int lpp = bpv_small_fused(tp);
    return !lpp ? 0 : (lpp >= 8 * nv ? 4 : 2);
}
hipError_t launch_bpv_msm(int tp, int n_bits, int m, int k, size_t nproofs, const affine_mem* pts, const affine_mem* table, int c, int W,
                          const tdig_t* digF, const fq* varS, xyzz* vartab, xyzz* fixpart, const int* bad, uint8_t* verdicts, hipStream_t st,
                          int wip, int nm_tab, int pre, hipStream_t helper, hipEvent_t ev_fork, hipEvent_t ev_join) {
    const int nm = n_bits * m;
    const int shiftW = nm_tab > nm ? (nm_tab - nm) * W : 0;
    if (const int lpp = bpv_small_fused(tp)) {  // small batch: table additions and ladders in one launch
        const unsigned blocks = (unsigned)((nproofs + (256 / (2 * lpp)) - 1) / (256 / (2 * lpp)));
#define BPV_SMALL(LPP)                                                                                                                         \
    hipLaunchKernelGGL((k_bpv_small<LPP>), dim3(blocks), dim3(256), 0, st, nm, m, k, nproofs, pts, table, c, W, digF, shiftW, varS, vartab, bad, \
                       verdicts, wip, pre)
        if (lpp == 16)
            BPV_SMALL(16);
        else if (lpp == 32)
            BPV_SMALL(32);
        else if (lpp == 64)
            BPV_SMALL(64);
        else
            BPV_SMALL(128);
#undef BPV_SMALL
        return hipGetLastError();
    }
    const int tf = bpv_fix_lanes(tp < 0 ? 16 : tp, nproofs, nm);
    unsigned fblocks = (unsigned)((nproofs + (256 / tf) - 1) / (256 / tf));
    // helper != nullptr: the fixed-generator kernel on the helper stream BESIDE the proof-specific kernel (one call alone leaves
    // room for both: 65 536 proofs 13.2 -> ~12 ms for the pair, 4096 proofs 1.64 -> ~1.0 ms); the sums meet in k_bpv_verdict
    hipStream_t sf = st;
    xyzz* varpart = nullptr;
    if (helper) {
        hipError_t e;
        if ((e = hipEventRecord(ev_fork, st)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(helper, ev_fork, 0)) != hipSuccess) return e;
        sf = helper;
        varpart = fixpart + nproofs;
    }
#define BPV_FIX(TP) hipLaunchKernelGGL((k_bpv_fix<TP>), dim3(fblocks), dim3(256), 0, sf, nm, nproofs, table, c, W, digF, bad, shiftW, fixpart)
    if (tf == 4)
        BPV_FIX(4);
    else if (tf == 8)
        BPV_FIX(8);
    else if (tf == 16)
        BPV_FIX(16);
    else if (tf == 32)
        BPV_FIX(32);
    else if (tf == 64)
        BPV_FIX(64);
    else
        BPV_FIX(2);
#undef BPV_FIX
    if (helper) {
        hipError_t e;
        if ((e = hipEventRecord(ev_join, helper)) != hipSuccess) return e;
    }
    auto finish = [&]() -> hipError_t {  // after the proof-specific kernel
        if (!helper) return hipGetLastError();
        hipError_t e;
        if ((e = hipStreamWaitEvent(st, ev_join, 0)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_bpv_verdict, dim3((unsigned)((nproofs + 255) / 256)), dim3(256), 0, st, m, k, nproofs, pts, fixpart, varpart, bad, verdicts,
                           wip);
        return hipGetLastError();
    };
    if (tp < 0) {
        const int lpp = -tp;
        unsigned sblocks = (unsigned)((nproofs * (size_t)lpp + 255) / 256);
#define BPV_SLICE(LPP) hipLaunchKernelGGL((k_bpv_msm_slice<LPP>), dim3(sblocks), dim3(256), 0, st, m, k, nproofs, pts, fixpart, varS, vartab, bad, verdicts, wip, varpart)
        if (lpp == 16)
            BPV_SLICE(16);
        else if (lpp == 32)
            BPV_SLICE(32);
        else if (lpp == 64)
            BPV_SLICE(64);
        else if (lpp == 128)
            BPV_SLICE(128);
        else
            BPV_SLICE(256);
#undef BPV_SLICE
        return finish();
    }
    unsigned blocks = (unsigned)((nproofs + (256 / tp) - 1) / (256 / tp));
    // vartab = [XYZZ construction slots | finished 64-byte tables | digits], each per lane
    affine_mem* varaff = reinterpret_cast<affine_mem*>(vartab + (size_t)blocks * 256 * bpv_ppl(tp) * ECMUL_SLOTS);
    int8_t* vardig = reinterpret_cast<int8_t*>(varaff + (size_t)blocks * 256 * bpv_ppl(tp) * ECMUL_NT);
#define BPV_LAUNCH(TP, PPL)                                                                                                                  \
    hipLaunchKernelGGL((k_bpv_msm<TP, PPL>), dim3(blocks), dim3(256), 0, st, n_bits, m, k, nproofs, pts, fixpart, varS, vartab, varaff, vardig, bad, \
                       verdicts, wip, varpart)
    if (tp == 2)
        BPV_LAUNCH(2, 8);
    else if (tp == 4)
        BPV_LAUNCH(4, 4);
    else if (tp == 8)
        BPV_LAUNCH(8, 2);
    else if (tp == 16)
        BPV_LAUNCH(16, 1);
    else if (tp == 32)
        BPV_LAUNCH(32, 1);
    else
        BPV_LAUNCH(64, 1);
#undef BPV_LAUNCH
    return finish();
}

}  // namespace bp

BP_OPCOUNT_READER(bp_opcount_verify)
