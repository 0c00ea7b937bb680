//! `extern "C"` view of `include/bpmsm.h`.  Points are 64 bytes (x || y big-endian, 64 zero bytes = identity),
//! scalars 32 bytes big-endian (reduced mod q by the library exactly like `Scalar::from(&BigInt)`).
//! Return codes: 0 = OK, BP_EINVAL where the reference asserts (src/proofs/range_proof.rs:556-562), BP_EPOINT where
//! `Point::from_bytes` would fail, BP_EHIP for runtime errors (`bp_last_error`).  A failed verification is a verdict
//! byte (enum order of src/lib.rs:30-35), never an error.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_double, c_int, c_long, c_void};

pub const BP_OK: c_int = 0;
pub const BP_EINVAL: c_int = -1;
pub const BP_EPOINT: c_int = -2;
pub const BP_EHIP: c_int = -3;
pub const BP_ERCCL: c_int = -4;
pub const BP_ENOMEM: c_int = -5;

pub const BP_VERDICT_OK: u8 = 0;
pub const BP_VERDICT_INNER_PRODUCT_ERROR: u8 = 1;
pub const BP_VERDICT_WEIGHTED_INNER_PROD_ERROR: u8 = 2;
pub const BP_VERDICT_RANGE_PROOF_ERROR: u8 = 3;
pub const BP_VERDICT_MALFORMED: u8 = 255;

pub const BP_MODE_VERIFY: c_int = 0;
pub const BP_MODE_FAST_VERIFY: c_int = 1;
pub const BP_MODE_AGGREGATED_VERIFY: c_int = 2;

#[repr(C)]
pub struct bp_ctx {
    _p: [u8; 0],
}
#[repr(C)]
pub struct bp_gens {
    _p: [u8; 0],
}
/// One job of `bp_msm_batch`: 2nm+2 generator scalars (nullable) plus job-specific points.
#[repr(C)]
pub struct bp_msm_job {
    pub gen_scalars_be: *const u8,
    pub extra_points_xy: *const u8,
    pub extra_scalars_be: *const u8,
    pub n_extra: usize,
}

extern "C" {
    pub fn bp_ctx_create(device: c_int, out: *mut *mut bp_ctx) -> c_int;
    pub fn bp_ctx_destroy(ctx: *mut bp_ctx);
    pub fn bp_last_error() -> *const c_char;
    pub fn bp_device_count() -> c_int;
    pub fn bp_ctx_sync(ctx: *mut bp_ctx) -> c_int;
    /// `hip_event`: a `hipEvent_t` of the caller (stream bridging for the asynchronous device entry points)
    pub fn bp_ctx_wait_event(ctx: *mut bp_ctx, hip_event: *mut c_void) -> c_int;
    pub fn bp_ctx_record_event(ctx: *mut bp_ctx, hip_event: *mut c_void) -> c_int;

    // every `map(mul).fold(add)` site: range_proof.rs:116-122, 236-244, 770-774; inner_product.rs:73-115, 336-338;
    // range_proof_wip.rs:249-251, 370-372, 598-602; weighted_inner_product.rs:109-154, 500-504
    pub fn bp_msm(ctx: *mut bp_ctx, points_xy: *const u8, scalars_be: *const u8, n: usize, init_xy: *const u8, out_xy: *mut u8) -> c_int;
    pub fn bp_msm_device(ctx: *mut bp_ctx, d_points_xy: *const c_void, d_scalars_be: *const c_void, n: usize, init_xy: *const u8,
                         out_xy: *mut u8) -> c_int;
    pub fn bp_msm_auto_window(n: usize) -> c_int;
    pub fn bp_msm_partial_bytes(c: c_int) -> usize;
    pub fn bp_msm_partial_device(ctx: *mut bp_ctx, d_points_xy: *const c_void, d_scalars_be: *const c_void, n: usize, c: c_int,
                                 partial_out: *mut u8) -> c_int;
    pub fn bp_msm_partial(ctx: *mut bp_ctx, points_xy: *const u8, scalars_be: *const u8, n: usize, c: c_int, partial_out: *mut u8) -> c_int;
    pub fn bp_msm_combine(ctx: *mut bp_ctx, partials: *const u8, nparts: usize, c: c_int, init_xy: *const u8, out_xy: *mut u8) -> c_int;
    pub fn bp_msm_partial_device_out(ctx: *mut bp_ctx, d_points_xy: *const c_void, d_scalars_be: *const c_void, n: usize, c: c_int,
                                     d_partial_out: *mut c_void) -> c_int;
    pub fn bp_msm_combine_device(ctx: *mut bp_ctx, d_partials: *const c_void, nparts: usize, c: c_int, init_xy: *const u8,
                                 out_xy: *mut u8) -> c_int;
    pub fn bp_msm_set_window(ctx: *mut bp_ctx, c: c_int);
    pub fn bp_msm_set_path(ctx: *mut bp_ctx, path: c_int);
    pub fn bp_msm_set_pipeline(ctx: *mut bp_ctx, mode: c_int);
    pub fn bp_proto_set_path(ctx: *mut bp_ctx, path: c_int);
    pub fn bp_verify_set_lanes(ctx: *mut bp_ctx, lanes: c_int);
    pub fn bp_verify_set_overlap(ctx: *mut bp_ctx, on: c_int);

    // g_vec / h_vec / G / H, uploaded once and expanded into the fixed-base window table
    pub fn bp_gens_create(ctx: *mut bp_ctx, g_xy: *const u8, h_xy: *const u8, nm: usize, G_xy: *const u8, H_xy: *const u8,
                          window_bits: c_int, out: *mut *mut bp_gens) -> c_int;
    pub fn bp_gens_destroy(gens: *mut bp_gens);
    pub fn bp_gens_table_bytes(gens: *const bp_gens) -> usize;
    pub fn bp_msm_batch(ctx: *mut bp_ctx, gens: *const bp_gens, jobs: *const bp_msm_job, njobs: usize, out_xy: *mut u8) -> c_int;

    // hi_tag (range_proof.rs:231-233), base folding (inner_product.rs:140-155), y*G / x_u*G (range_proof.rs:126, 219, 588)
    pub fn bp_scalar_mul_batch(ctx: *mut bp_ctx, points_xy: *const u8, scalars_be: *const u8, n: usize, out_xy: *mut u8) -> c_int;
    pub fn bp_fold_bases(ctx: *mut bp_ctx, lo_xy: *const u8, hi_xy: *const u8, n_half: usize, s_lo: *const u8, s_hi: *const u8,
                         out_xy: *mut u8) -> c_int;
    pub fn bp_fixed_base_mul_batch(ctx: *mut bp_ctx, scalars_be: *const u8, n: usize, out_xy: *mut u8) -> c_int;

    // generate_random_point / the tests' recipe / StatementRP::generate_bases (range_proof.rs:785-843, range_proof_wip.rs:48-91)
    pub fn bp_generate_bases(kind: c_int, seed: u64, n_bits: usize, m: usize, g_xy: *mut u8, h_xy: *mut u8, G_xy: *mut u8,
                             H_xy: *mut u8) -> c_int;
    pub fn bp_generate_bases_device(ctx: *mut bp_ctx, kind: c_int, seed: u64, n_bits: usize, m: usize, g_xy: *mut u8, h_xy: *mut u8,
                                    G_xy: *mut u8, H_xy: *mut u8) -> c_int;

    // RangeProof::{prove, verify, fast_verify, aggregated_verify} (range_proof.rs:45-782)
    pub fn bp_rangeproof_prove(ctx: *mut bp_ctx, gens: *const bp_gens, n_bits: c_int, secrets_be: *const u8, blindings_be: *const u8,
                               m: c_int, tape: *const u8, tape_len: usize, proof_out: *mut u8, proof_cap: usize,
                               proof_len: *mut usize) -> c_int;
    pub fn bp_rangeproof_prove_batch(ctx: *mut bp_ctx, gens: *const bp_gens, n_bits: c_int, m: c_int, nproofs: usize, secrets_be: *const u8,
                                     blindings_be: *const u8, tapes: *const u8, tape_len: usize, proofs_out: *mut u8, proof_len: usize,
                                     status: *mut c_int) -> c_int;
    pub fn bp_wip_rangeproof_prove_batch(ctx: *mut bp_ctx, gens: *const bp_gens, n_bits: c_int, m: c_int, nproofs: usize,
                                         secrets_be: *const u8, blindings_be: *const u8, tapes: *const u8, tape_len: usize,
                                         proofs_out: *mut u8, proof_len: usize, status: *mut c_int) -> c_int;
    pub fn bp_rangeproof_verify(ctx: *mut bp_ctx, gens: *const bp_gens, n_bits: c_int, m: c_int, proof: *const u8, proof_len: usize,
                                coms_xy: *const u8, mode: c_int, verdict: *mut u8) -> c_int;
    pub fn bp_rangeproof_verify_batch(ctx: *mut bp_ctx, gens: *const bp_gens, n_bits: c_int, m: c_int, proofs: *const u8,
                                      proof_len: usize, coms_xy: *const u8, nproofs: usize, verdicts: *mut u8) -> c_int;
    pub fn bp_rangeproof_verify_batch_device(ctx: *mut bp_ctx, gens: *const bp_gens, n_bits: c_int, m: c_int, d_proofs: *const c_void,
                                             proof_len: usize, d_coms_xy: *const c_void, nproofs: usize,
                                             d_verdicts: *mut c_void) -> c_int;
    // aggregated_verify x N with a different m per proof (range_proof.rs:537-782, range_proof_wip.rs:388-609)
    pub fn bp_rangeproof_verify_batch_mixed(ctx: *mut bp_ctx, gens: *const bp_gens, n_bits: c_int, proofs: *const u8, proof_off: *const usize,
                                            coms_xy: *const u8, m_per_proof: *const u32, nproofs: usize, verdicts: *mut u8) -> c_int;
    pub fn bp_wip_rangeproof_verify_batch_mixed(ctx: *mut bp_ctx, gens: *const bp_gens, n_bits: c_int, proofs: *const u8,
                                                proof_off: *const usize, coms_xy: *const u8, m_per_proof: *const u32, nproofs: usize,
                                                verdicts: *mut u8) -> c_int;
    // no counterpart in the reference: the same call for callers that cannot allocate (fixed 64 MiB device scratch per GPU)
    pub fn bp_rangeproof_verify_batch_device_noalloc(ctx: *mut bp_ctx, gens: *const bp_gens, n_bits: c_int, m: c_int,
                                                     d_proofs: *const c_void, proof_len: usize, d_coms_xy: *const c_void, nproofs: usize,
                                                     d_verdicts: *mut c_void) -> c_int;
    pub fn bp_noalloc_scratch_bytes() -> usize;
    pub fn bp_rangeproof_verify_batch_rlc(ctx: *mut bp_ctx, gens: *const bp_gens, n_bits: c_int, m: c_int, proofs: *const u8,
                                          proof_len: usize, coms_xy: *const u8, nproofs: usize, seed: *const u8, verdicts: *mut u8,
                                          used_fallback: *mut c_int) -> c_int;
    pub fn bp_rangeproof_verify_batch_rlc_device(ctx: *mut bp_ctx, gens: *const bp_gens, n_bits: c_int, m: c_int,
                                                 d_proofs: *const c_void, proof_len: usize, d_coms_xy: *const c_void, nproofs: usize,
                                                 seed: *const u8, d_verdicts: *mut c_void, used_fallback: *mut c_int) -> c_int;

    // InnerProductArg::{prove, verify, fast_verify} (inner_product.rs:36, 171, 253) and WeightedInnerProdArg::{prove, verify,
    // fast_verify} (weighted_inner_product.rs:47, 270, 386) over caller-supplied bases
    pub fn bp_ipa_prove(ctx: *mut bp_ctx, g_xy: *const u8, h_xy: *const u8, n: usize, ux_xy: *const u8, p_xy: *const u8, a_be: *const u8,
                        b_be: *const u8, arg_out: *mut u8, arg_cap: usize, arg_len: *mut usize) -> c_int;
    pub fn bp_ipa_verify(ctx: *mut bp_ctx, g_xy: *const u8, hi_tag_xy: *const u8, n: usize, ux_xy: *const u8, p_xy: *const u8,
                         arg: *const u8, arg_len: usize, mode: c_int, verdict: *mut u8) -> c_int;
    pub fn bp_wipa_prove(ctx: *mut bp_ctx, g_xy: *const u8, h_xy: *const u8, n: usize, g_pt: *const u8, h_pt: *const u8, p_xy: *const u8,
                         a_be: *const u8, b_be: *const u8, alpha_be: *const u8, y_be: *const u8, tape: *const u8, tape_len: usize,
                         arg_out: *mut u8, arg_cap: usize, arg_len: *mut usize) -> c_int;
    pub fn bp_wipa_verify(ctx: *mut bp_ctx, g_xy: *const u8, hi_tag_xy: *const u8, n: usize, g_pt: *const u8, h_pt: *const u8,
                          p_xy: *const u8, y_be: *const u8, arg: *const u8, arg_len: usize, mode: c_int, verdict: *mut u8) -> c_int;

    // serde wire formats of RangeProof / RangeProofWIP <-> canonical proof bytes (kind 0 / 1; format 0 = JSON, 1 = bincode)
    pub fn bp_proof_to_serde(kind: c_int, format: c_int, proof: *const u8, proof_len: usize, out: *mut u8, cap: usize,
                             out_len: *mut usize) -> c_int;
    pub fn bp_proof_from_serde(kind: c_int, format: c_int, input: *const u8, in_len: usize, proof_out: *mut u8, cap: usize,
                               proof_len: *mut usize) -> c_int;

    // RangeProofWIP::{prove, verify, aggregated_verify} (range_proof_wip.rs:101-609)
    pub fn bp_wip_rangeproof_prove(ctx: *mut bp_ctx, gens: *const bp_gens, n_bits: c_int, secrets_be: *const u8, blindings_be: *const u8,
                                   m: c_int, tape: *const u8, tape_len: usize, proof_out: *mut u8, proof_cap: usize,
                                   proof_len: *mut usize) -> c_int;
    pub fn bp_wip_rangeproof_verify(ctx: *mut bp_ctx, gens: *const bp_gens, n_bits: c_int, m: c_int, proof: *const u8,
                                    proof_len: usize, coms_xy: *const u8, mode: c_int, verdict: *mut u8) -> c_int;
    pub fn bp_wip_rangeproof_verify_batch(ctx: *mut bp_ctx, gens: *const bp_gens, n_bits: c_int, m: c_int, proofs: *const u8,
                                          proof_len: usize, coms_xy: *const u8, nproofs: usize, verdicts: *mut u8) -> c_int;
    pub fn bp_wip_rangeproof_verify_batch_device(ctx: *mut bp_ctx, gens: *const bp_gens, n_bits: c_int, m: c_int,
                                                 d_proofs: *const c_void, proof_len: usize, d_coms_xy: *const c_void, nproofs: usize,
                                                 d_verdicts: *mut c_void) -> c_int;
    pub fn bp_wip_rangeproof_verify_batch_rlc(ctx: *mut bp_ctx, gens: *const bp_gens, n_bits: c_int, m: c_int, proofs: *const u8,
                                              proof_len: usize, coms_xy: *const u8, nproofs: usize, seed: *const u8, verdicts: *mut u8,
                                              used_fallback: *mut c_int) -> c_int;
    pub fn bp_wip_rangeproof_verify_batch_rlc_device(ctx: *mut bp_ctx, gens: *const bp_gens, n_bits: c_int, m: c_int,
                                                     d_proofs: *const c_void, proof_len: usize, d_coms_xy: *const c_void,
                                                     nproofs: usize, seed: *const u8, d_verdicts: *mut c_void,
                                                     used_fallback: *mut c_int) -> c_int;

    // per-kernel timing of the batch verifier (HIP events on the library's stream)
    pub fn bp_profile_enable(ctx: *mut bp_ctx, on: c_int);
    pub fn bp_profile_reset(ctx: *mut bp_ctx);
    pub fn bp_profile_get(ctx: *mut bp_ctx, kernel_id: c_int, total_ms: *mut c_double, launches: *mut c_long) -> c_int;
}
