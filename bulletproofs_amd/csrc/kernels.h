// Host-callable launchers of the HIP kernels (defined in the .hip files of this directory).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "ec.h"
#include "fq.h"

#include <stdlib.h>

namespace bp {

// Tuning knobs (BP_MSM_*, BP_SRT2_HB, BP_FIX_LANES, BP_VERIFY_SMALL_SPLIT, BP_PROVE_CHUNK: experiment sweeps under scripts/)
// are read from the environment only in -DBP_TUNING builds (make EXTRA=-DBP_TUNING); the product build never reads it.
#ifdef BP_TUNING
static inline const char* bp_tune(const char* name) { return getenv(name); }
#else
static inline const char* bp_tune(const char*) { return nullptr; }
#endif

// signed window digit of a fixed-base table term (tables go up to 20-bit windows: |digit| <= 2^19)
typedef int32_t tdig_t;

// msm_kernels.hip
size_t msm_slice_lds_bytes(int c, int M, int T);
hipError_t launch_ingest_points(const uint8_t* in_be, affine_mem* out, size_t n, int* err, hipStream_t st);
hipError_t launch_export_points(const affine_mem* in, uint8_t* out_be, size_t n, hipStream_t st);
hipError_t launch_scalars_to_digits(const uint8_t* scalars_be, size_t n, size_t npad, int c, int W, int16_t* digits, hipStream_t st);
hipError_t launch_glv_expand(const affine_mem* in, affine_mem* lam, size_t n, hipStream_t st);
hipError_t launch_glv_digits(const uint8_t* scalars_be, size_t n, size_t npad, int c, int W, int16_t* digits, hipStream_t st);
hipError_t launch_msm_slice(const affine_mem* pts, const affine_mem* lam, size_t n1, const int16_t* digits, size_t n, size_t npad, int M, int c,
                            int W, int S, int T, xyzz* partial, hipStream_t st);
hipError_t launch_msm_reduce_slices(const xyzz* partial, int S, int W, xyzz* wsum, hipStream_t st);
hipError_t launch_msm_reduce_parts(const xyzz* parts, int G, int W, xyzz* wsum, hipStream_t st);
size_t msm_tiny_max();
size_t msm_tiny_scratch_bytes();
hipError_t launch_msm_tiny(const affine_mem* pts, const uint8_t* scalars_be, size_t n, const affine_mem* init, xyzz* tabs, affine_mem* out_limbs,
                           uint8_t* out_be, hipStream_t st);
hipError_t launch_or_flag(int* dst, const int* src, hipStream_t st);
hipError_t launch_msm_horner(const xyzz* wsum, int W, int c, const affine_mem* init, affine_mem* out_limbs, uint8_t* out_be, hipStream_t st,
                             xyzz* out_raw = nullptr, const int* err_in = nullptr);


// msm_sorted.hip: large-N path (global counting sort by bucket, balanced segment accumulation)
struct SortedPlan {
    int c, W, NB, CH, nchunks, L, nseg, lo_bits;
    int Lmin, pstride;  // shortest segment a window group may use (launch_msm_sorted) and the stride of Pseg it implies
    int ncu;  // compute units of the device (0 = unknown)
    int sort2 = 0;  // 1 = two-pass sort at any size (bp_msm_set_path(ctx, 3): tests), 0 = from 2^15 virtual points
    int weighted = 0;  // 1 = one-group path: window w's workgroup also multiplies its sum by 2^(c w) (full MSMs only, see k_srt_window)
    size_t npad;
};
struct SortedBufs {
    uint32_t *cnt, *bstart;
    uint2* ent;
    uint2* tmp;  // partition-sorted intermediate of the two-pass sort ([W][npad] like ent); nullptr = one-pass sort only
    xyzz *Bk, *Pseg, *RC;
};
size_t srt2_cnt_words(const SortedPlan& pl, size_t n2);  // 32-bit words of `cnt` the two-pass sort needs
// side streams + events of the grouped pipeline (see launch_msm_sorted): st2 sorts, st3 bucket reductions, st4 Horner chain
struct SortedStreams {
    static const int kMaxGroups = 8;
    hipStream_t st2, st3, st4;
    hipEvent_t ev_acc[kMaxGroups], ev_red[kMaxGroups], ev_sorted[kMaxGroups], ev_join, ev_fork, ev_start, ev_lam;
};
hipError_t launch_msm_sorted(const affine_mem* pts, const affine_mem* lam, size_t n1, const int16_t* digits, size_t n2, const SortedPlan& pl,
                             const SortedBufs& b, xyzz* wsum, hipStream_t st, const SortedStreams* ss, const int* gsz, int ng, int* W_eff,
                             int* c_eff);

// gens_kernels.hip
hipError_t launch_gens_bases(const affine_mem* pts, int np, int c, int W, xyzz* bases, hipStream_t st);
hipError_t launch_xyzz_to_affine(const xyzz* in, affine_mem* out, size_t n, hipStream_t st);
hipError_t launch_gens_multiples(const affine_mem* basesA, size_t nidx, int NB, xyzz* tmp, hipStream_t st);
hipError_t launch_gens_normalize(const xyzz* tmp, size_t n, affine_mem* table, hipStream_t st);

// gens_derive.hip: the reference's synthetic generator recipe, one lane per generator (g_vec, h_vec, G, H order)
hipError_t launch_gens_derive(int kind, uint64_t seed, uint64_t n_bits, size_t nm, affine_mem* out, hipStream_t st);

// verify_kernels.hip
hipError_t launch_bpv_decompress(const uint8_t* proofs, size_t proof_len, const uint8_t* coms_xy, int k, int m, size_t nproofs, affine_mem* pts,
                                 int* bad, hipStream_t st);
// verify_scalars.hip
size_t bpv_chal_count(int k);
size_t bpv_cst_count(int k, int m);
hipError_t launch_bpv_scalar_pipeline(const uint8_t* proofs, size_t proof_len, int n_bits, int m, int k, size_t nproofs, const affine_mem* pts,
                                      const affine_mem* gens, const affine_mem* table, int c, int W, tdig_t* digF, fqm* scF, fq* varS, fq* chal,
                                      fq* cst, xyzz* fixres, affine_mem* fixaff, int* bad, hipStream_t st, int ibase, int nm_tab,
                                      xyzz* pretab = nullptr, int lpp = 0, int nparts = 2);  // pretab/lpp/nparts: small-batch window tables (bpv_small_fused, bpv_small_parts)
// rlc_kernels.hip: random-linear-combination batch verification (one MSM for the whole batch)
hipError_t launch_rlc_weights(const uint8_t* seed32, size_t nproofs, const int* bad, fqm* rho, hipStream_t st);
hipError_t launch_rlc_var_scalars(int k, int m, size_t nproofs, const fqm* rho, const fq* varS, uint8_t* sc_be, hipStream_t st, int wip = 0);
size_t rlc_colsum_chunks(size_t nproofs);
hipError_t launch_rlc_colsum(int np, size_t nproofs, const fqm* rho, const fqm* scF, const int* bad, fqm* part, uint8_t* out_be, hipStream_t st);
hipError_t launch_rlc_verdicts(size_t nproofs, const uint8_t* result_be, const int* bad, uint8_t* verdicts, int* all_ok, hipStream_t st);
hipError_t launch_bpv_msm(int tp, int n_bits, int m, int k, size_t nproofs, const affine_mem* pts, const affine_mem* table, int c, int W,
                          const tdig_t* digF, const fq* varS, xyzz* vartab, xyzz* fixpart, const int* bad, uint8_t* verdicts, hipStream_t st,
                          int wip = 0, int nm_tab = 0, int pre = 0,  // pre: vartab already holds the slices' window tables
                          hipStream_t helper = nullptr, hipEvent_t ev_fork = nullptr, hipEvent_t ev_join = nullptr);  // helper: stream for k_bpv_fix
size_t bpv_msm_fixpart_bytes(size_t nproofs);
// prove_kernels.hip: scalar side of the batch prover RangeProof::prove x N (vectors over [proof][i]) and its table MSMs
struct pv_ent {  // one term of a table MSM: table slot (row W + window) and the signed window digit
    uint32_t slot;
    int32_t d;
};
hipError_t launch_pv_init(size_t nproofs, int nm, int c, int W, const uint8_t* tapes, size_t tape_len, const uint8_t* bits, fqm* SL, fqm* SR,
                          pv_ent* ent, size_t stride, hipStream_t st);
hipError_t launch_pv_term(size_t nproofs, int c, int W, int row, const uint8_t* scalars_be, pv_ent* ent, size_t stride, hipStream_t st);
hipError_t launch_pv_poly(size_t nproofs, int n_bits, int m, int c, int W, const uint8_t* tapes, size_t tape_len, const uint8_t* yz_be,
                          const uint8_t* bits, const fqm* SL, const fqm* SR, fqm* yi, fqm* zpow, fqm* yinv, pv_ent* ent, size_t stride,
                          hipStream_t st);
hipError_t launch_pv_lr(size_t nproofs, int n_bits, int m, const uint8_t* tapes, size_t tape_len, const uint8_t* yz_be, const uint8_t* x_be,
                        const uint8_t* blindings_be, const uint8_t* bits, const fqm* SL, const fqm* SR, const fqm* yi, const fqm* zpow,
                        const fqm* yinv, fqm* a, fqm* b, fqm* cG, fqm* cH, uint8_t* out_be, hipStream_t st);
hipError_t launch_pv_round(size_t nproofs, int nm, int n, int c, int W, const uint8_t* xu_be, const fqm* a, const fqm* b, const fqm* cG,
                           const fqm* cH, pv_ent* ent, size_t stride, hipStream_t st);
hipError_t launch_pv_fold(size_t nproofs, int nm, int n, const uint8_t* xj_be, fqm* a, fqm* b, fqm* cG, fqm* cH, hipStream_t st);
#define PV_MSM_MAXPARTS 8  // workgroups per job of k_pv_msm at most: `partial` holds njobs * PV_MSM_MAXPARTS sums
hipError_t launch_pv_msm(size_t njobs, const affine_mem* table, int c, const pv_ent* ent, size_t stride, int cnt0, int cnt1, xyzz* partial,
                         xyzz* out, hipStream_t st);
// Bulletproofs+ batch prover (RangeProofWIP::prove x N)
hipError_t launch_pw_init(size_t nproofs, int nm, int c, int W, const uint8_t* tapes, size_t tape_len, const uint8_t* bits, pv_ent* ent,
                          size_t stride, hipStream_t st);
hipError_t launch_pw_setup(size_t nproofs, int n_bits, int m, const uint8_t* tapes, size_t tape_len, const uint8_t* yz_be,
                           const uint8_t* blindings_be, const uint8_t* bits, fqm* yp, fqm* a, fqm* b, fqm* cG, fqm* cH, fqm* alpha_hat, fqm* yinv,
                           hipStream_t st);
hipError_t launch_pw_round(size_t nproofs, int nm, int n, int j, int c, int W, const uint8_t* tapes, size_t tape_len, const fqm* yp, const fqm* yinv,
                           const fqm* a, const fqm* b, const fqm* cG, const fqm* cH, pv_ent* ent, size_t stride, hipStream_t st);
hipError_t launch_pw_fold(size_t nproofs, int nm, int n, int j, const uint8_t* tapes, size_t tape_len, const uint8_t* xj_be, const fqm* yp,
                          const fqm* yinv, fqm* a, fqm* b, fqm* cG, fqm* cH, fqm* alpha_hat, hipStream_t st);
hipError_t launch_pw_last(size_t nproofs, int nm, int k, int c, int W, const uint8_t* tapes, size_t tape_len, const uint8_t* yz_be, const fqm* a,
                          const fqm* b, const fqm* cG, const fqm* cH, pv_ent* ent, size_t stride, hipStream_t st);
hipError_t launch_pw_finish(size_t nproofs, int nm, int k, const uint8_t* tapes, size_t tape_len, const uint8_t* e_be, const fqm* a, const fqm* b,
                            const fqm* alpha_hat, uint8_t* out_be, hipStream_t st);
hipError_t launch_pv_final(size_t nproofs, int nm, const fqm* a, const fqm* b, uint8_t* out_be, hipStream_t st);
// lanes per proof (> 0) when a batch with tp lanes takes the one-launch small-batch MSM (k_bpv_small) and its window tables
// are built beside the constants kernel (k_bpv_consts_pre); 0 otherwise
int bpv_small_fused(int tp);
int bpv_small_parts(int tp, int nv);  // window ranges per GLV half of the fused small-batch kernels (0: not fused)
// Bulletproofs+ twins (RangeProofWIP::aggregated_verify): proof-point order b_tag, A, L[k], R[k], a_tag, V[m]
hipError_t launch_wipv_decompress(const uint8_t* proofs, size_t proof_len, const uint8_t* coms_xy, int k, int m, size_t nproofs, affine_mem* pts,
                                  int* bad, hipStream_t st);
hipError_t launch_wipv_scalar_pipeline(const uint8_t* proofs, size_t proof_len, int n_bits, int m, int k, size_t nproofs, const affine_mem* pts,
                                       const affine_mem* gens, const affine_mem* table, int c, int W, tdig_t* digF, fqm* scF, fq* varS, fq* chal,
                                       fq* cst, xyzz* fixres, affine_mem* fixaff, int* bad, hipStream_t st, int ibase, int nm_tab,
                                       xyzz* pretab = nullptr, int lpp = 0, int nparts = 2);
size_t bpv_msm_vartab_bytes(int tp, size_t nproofs);
int bpv_slice_lanes(int nv, int nparts = 2);


// msm_batch.hip
hipError_t launch_mb_digits(const uint8_t* scalars_be, size_t total, int c, int W, tdig_t* digits, hipStream_t st, int mont = 0);
size_t mb_vartab_bytes(size_t njobs);
hipError_t launch_mb_msm(size_t njobs, int np, const affine_mem* table, int c, int W, const tdig_t* digF, const uint8_t* has_gen,
                         const affine_mem* xpts, const uint8_t* xsc_be, const uint32_t* xoff, xyzz* vartab, xyzz* out, hipStream_t st);

// fbmsm.hip: few jobs over the generator table, whole-chip parallel (prover / single-proof verifier)
size_t fb_vartab_bytes(size_t njobs, int nxblk);
hipError_t launch_fb_msm(size_t njobs, const affine_mem* table, int W, int NB, const tdig_t* dig, const uint32_t* gidx, const uint32_t* toff,
                         const affine_mem* xpts, const uint8_t* xsc_be, const uint32_t* xoff, int nblk, int nxblk, xyzz* vartab, xyzz* partial,
                         uint8_t* out_be, hipStream_t st, int pre = 0, xyzz* out_raw = nullptr);
#define FB_PRE_PARTS 3  // window ranges per GLV half of the tables k_fb_pretab builds (launch_fb_msm's `pre`)
struct FbPreJobs {  // point ranges of up to 8 jobs: job j owns points off[j] .. off[j + 1)
    uint32_t off[9];
};
hipError_t launch_fb_pretab(const uint8_t* xbe, const FbPreJobs& jobs, int nj, int nxblk, xyzz* vartab, hipStream_t st);

// ecbatch_kernels.hip
size_t ecbatch_vartab_bytes(size_t n);
hipError_t launch_scalar_mul_batch(const affine_mem* pts, const uint8_t* scalars_be, size_t n, affine_mem* out, xyzz* vartab, hipStream_t st);
hipError_t launch_fold_bases(const affine_mem* lo, const affine_mem* hi, size_t n_half, const uint8_t* s_lo_be, const uint8_t* s_hi_be,
                             affine_mem* out, xyzz* vartab, hipStream_t st);

}  // namespace bp
