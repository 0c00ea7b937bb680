/* bpmsm — C ABI of the MI355X-native secp256k1 MSM hot path of ZenGo-X/bulletproofs.
 *
 * The reference has no FFI: its boundary is the operator API of the external crate curv-kzen 0.10
 * (`&Point * &Scalar`, `Point + Point`, `BigInt::mod_*`, `DigestExt`) used through the idiom
 *     (0..n).map(|i| &points[i] * &Scalar::from(&scalars[i])).fold(init, |acc, x| acc + x)
 * Each entry point below names the reference call sites (file:line under /root/reference) whose
 * computation it replaces; INTEGRATION.md shows the Rust `-sys` binding a maintainer would add.
 *
 * Conventions
 *   point   : 64 bytes, x || y, each 32-byte big-endian, canonical (< p); 64 zero bytes = infinity.
 *   scalar  : 32 bytes big-endian; any 256-bit value is accepted and reduced mod q exactly like
 *             `Scalar::<Secp256k1>::from(&BigInt)`.
 *   compressed point (inside proof encodings): 33-byte SEC1 (curv `to_bytes(true)`).
 *   Ownership: the caller owns every buffer; nothing is retained after return except by bp_gens
 *   (which copies).  Outputs are written only on success.  All pointers are HOST pointers unless a
 *   function says "device".
 *   Return value: 0 = OK, negative = usage/runtime error (below).  A failed verification is NOT an
 *   error: it is a verdict (mirrors `Result<(), Errors>`, src/lib.rs:30-35).
 *   Threading: a bp_ctx serialises its calls on one HIP stream; use one ctx per thread/GPU.
 *   bp_gens is immutable and may be shared by contexts on the same device.
 */
#ifndef BPMSM_H
#define BPMSM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BP_OK 0
#define BP_EINVAL (-1)  /* bad sizes / non-power-of-two nm: where the reference panics (range_proof.rs:556-562) */
#define BP_EPOINT (-2)  /* an input point is not on the curve (curv Point::from_bytes would fail)  */
#define BP_EHIP (-3)    /* HIP runtime error; bp_last_error() has the text                          */
#define BP_ERCCL (-4)   /* reserved for the multi-GPU layer                                         */
#define BP_ENOMEM (-5)

/* Verdict bytes written by the verifiers: enum order of `Errors` in src/lib.rs:30-35. */
#define BP_VERDICT_OK 0
#define BP_VERDICT_INNER_PRODUCT_ERROR 1
#define BP_VERDICT_WEIGHTED_INNER_PROD_ERROR 2
#define BP_VERDICT_RANGE_PROOF_ERROR 3
#define BP_VERDICT_MALFORMED 255 /* encoding the reference could not deserialize (bad point / scalar >= q) */

typedef struct bp_ctx bp_ctx;
typedef struct bp_gens bp_gens;

/* Library / device management (no reference counterpart: the reference is a single-threaded CPU crate). */
int bp_ctx_create(int device, bp_ctx** out);
void bp_ctx_destroy(bp_ctx* ctx);
const char* bp_last_error(void);
int bp_device_count(void);

/* bp_msm — one multi-scalar multiplication  out = init + sum_i scalars[i] * points[i].
 * Replaces every `map(mul).fold(add)` site: range_proof.rs:116-122 (S), 236-244 (P), 770-774
 * (aggregated_verify); inner_product.rs:73-115 (L, R), 336-338 (fast_verify);
 * range_proof_wip.rs:249-251, 370-372, 598-602; weighted_inner_product.rs:109-154, 500-504.
 * init_xy may be NULL (= infinity).  Zero scalars and infinity points are legal (the reference's
 * zero-skip at inner_product.rs:74 is the same group element).  n may be 0. */
int bp_msm(bp_ctx* ctx, const uint8_t* points_xy, const uint8_t* scalars_be, size_t n, const uint8_t* init_xy, uint8_t out_xy[64]);

/* Same computation with inputs already resident in device memory (HBM): d_points_xy / d_scalars_be
 * are DEVICE pointers to the same byte layouts; out_xy is a host pointer.  This is the entry the
 * throughput benchmark times (inputs resident, see bench.py). */
int bp_msm_device(bp_ctx* ctx, const void* d_points_xy, const void* d_scalars_be, size_t n, const uint8_t* init_xy, uint8_t out_xy[64]);

/* Testing / tuning knob: force the Pippenger window width (0 = automatic). */
void bp_msm_set_window(bp_ctx* ctx, int c);

#ifdef __cplusplus
}
#endif
#endif /* BPMSM_H */
