// C ABI: generator tables (bp_gens) and the batched range-proof verifier.  Host plumbing only.
#include <algorithm>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/bpmsm.h"
#include "ctx.h"
#include "kernels.h"

using namespace bp;

#define HIPCHK(expr, where)        \
    do {                           \
        hipError_t e__ = (expr);   \
        if (e__ != hipSuccess) {   \
            set_error(where, e__); \
            return BP_EHIP;        \
        }                          \
    } while (0)

#define BP_MAX_TABLE_WINDOW 20
// XYZZ scratch of the table build: rows are built in groups of at most this many bytes (round 2 built all rows at once: 2.25x
// the table, 119 GiB beside the 52.8 GiB table of n=64 m=1; a group of 8 GiB still gives every SIMD a chain)
#define BP_GENS_TMP_BYTES ((size_t)8 << 30)
#define BP_NOALLOC_SCRATCH_BYTES (64u << 20)  // 64 MiB per GPU, see bp_rangeproof_verify_batch_device_noalloc
static int pick_table_window(int np) {
    // widest c in [4, 20] whose table (np * W * 2^(c-1) * 64 B) stays <= 64 GiB of the 288 GB HBM and whose build
    // (table + the XYZZ scratch of one group of rows, at most BP_GENS_TMP_BYTES, freed afterwards) fits in 80 % of the
    // memory that is free right now.
    // n=64 m=1: c = 20, 52.8 GiB, 13 gathers per generator term (c = 16: 4.3 GiB, 17 gathers); measured with a
    // 68 GiB table (scripts/experiments/bigtable_test.py, profiles/r02_bigtable_gather.txt): a table addition costs
    // 10 % more than from a 5.5 GiB table (74.9 -> 82.1 ps) while the wider window removes 4 of 17 additions per
    // generator term (-24 %), so HBM capacity still buys additions.
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)8 << 30;
    int best = 4;
    for (int c = 4; c <= BP_MAX_TABLE_WINDOW; c++) {
        size_t bytes = (size_t)np * (256 / c + 1) * ((size_t)1 << (c - 1)) * 64;
        double need = (double)bytes + (double)std::min<size_t>(bytes / 64 * 144, BP_GENS_TMP_BYTES) + (double)((size_t)2 << 30);
        if (bytes <= ((size_t)64 << 30) && need <= 0.8 * (double)free_b) best = c;
    }
    return best;
}

// secp256k1 generator, x || y big-endian
static const uint8_t kSecpGenerator[64] = {
    0x79, 0xBE, 0x66, 0x7E, 0xF9, 0xDC, 0xBB, 0xAC, 0x55, 0xA0, 0x62, 0x95, 0xCE, 0x87, 0x0B, 0x07, 0x02, 0x9B, 0xFC, 0xDB, 0x2D, 0xCE,
    0x28, 0xD9, 0x59, 0xF2, 0x81, 0x5B, 0x16, 0xF8, 0x17, 0x98, 0x48, 0x3A, 0xDA, 0x77, 0x26, 0xA3, 0xC4, 0x65, 0x5D, 0xA4, 0xFB, 0xFC,
    0x0E, 0x11, 0x08, 0xA8, 0xFD, 0x17, 0xB4, 0x48, 0xA6, 0x85, 0x54, 0x19, 0x9C, 0x47, 0xD0, 0x8F, 0xFB, 0x10, 0xD4, 0xB8};

extern "C" {

static int gens_create_locked(bp_ctx* ctx, const uint8_t* g_xy, const uint8_t* h_xy, size_t nm, const uint8_t G_xy[64], const uint8_t H_xy[64], int c,
                              bp_gens** out, bool* oom);

int bp_gens_create(bp_ctx* ctx, const uint8_t* g_xy, const uint8_t* h_xy, size_t nm, const uint8_t G_xy[64], const uint8_t H_xy[64],
                   int window_bits, bp_gens** out) {
    if (!ctx || !out || !g_xy || !h_xy || !G_xy || !H_xy || nm == 0 || nm > (1u << 20)) return BP_EINVAL;
    *out = nullptr;
    CtxLock lk(ctx);
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    const int np = (int)(2 * nm + 2) + (memcmp(G_xy, kSecpGenerator, 64) != 0 ? 1 : 0);  // table rows (see bp_gens::ibase)
    int c = window_bits ? window_bits : pick_table_window(np);
    if (c < 2 || c > BP_MAX_TABLE_WINDOW) return BP_EINVAL;
    // automatic window: the free-memory reading can be stale (another process building its table on the same GPU), so an
    // out-of-memory build is retried with a narrower window instead of failing
    for (;;) {
        bool oom = false;
        int rc = gens_create_locked(ctx, g_xy, h_xy, nm, G_xy, H_xy, c, out, &oom);
        if (rc == BP_OK || window_bits || !oom || c <= 8) return rc;
        (void)hipGetLastError();
        c -= 2;
    }
}

static int gens_create_locked(bp_ctx* ctx, const uint8_t* g_xy, const uint8_t* h_xy, size_t nm, const uint8_t G_xy[64], const uint8_t H_xy[64], int c,
                              bp_gens** out, bool* oom) {
    hipStream_t st = ctx->stream;
    const bool own_base = memcmp(G_xy, kSecpGenerator, 64) != 0;
    int np = (int)(2 * nm + 2) + (own_base ? 1 : 0);  // table rows: g_vec, h_vec, G, H (+ the secp256k1 generator)
    int W = 256 / c + 1, NB = 1 << (c - 1);
    bp_gens* g = new bp_gens();
    g->device = ctx->device;
    g->nm = nm;
    g->np = (int)(2 * nm + 2);
    g->ibase = own_base ? (int)(2 * nm + 2) : (int)(2 * nm);
    g->nrows = np;
    g->c = c;
    g->W = W;
    g->NB = NB;
    size_t nidx = (size_t)np * W, nent = nidx * NB;
    g->table_bytes = nent * sizeof(affine_mem);
    uint8_t* d_be = nullptr;
    xyzz *d_bases = nullptr, *d_tmp = nullptr;
    affine_mem* d_basesA = nullptr;
    int rc = BP_OK;
    std::vector<uint8_t> host((size_t)np * 64);
    memcpy(host.data(), g_xy, nm * 64);
    memcpy(host.data() + nm * 64, h_xy, nm * 64);
    memcpy(host.data() + 2 * nm * 64, G_xy, 64);
    memcpy(host.data() + (2 * nm + 1) * 64, H_xy, 64);
    if (own_base) memcpy(host.data() + (2 * nm + 2) * 64, kSecpGenerator, 64);
    for (int i = 0; i < np && rc == BP_OK; i++) {
        bool allz = true;
        for (int j = 0; j < 64; j++) allz = allz && host[(size_t)i * 64 + j] == 0;
        if (allz) {
            set_error_msg("generator is the point at infinity");
            rc = BP_EPOINT;
        }
    }
    hipError_t e = hipSuccess;
#define GCHK(expr)                          \
    if (rc == BP_OK) {                      \
        e = (expr);                         \
        if (e != hipSuccess) {              \
            set_error("bp_gens_create", e); \
            if (e == hipErrorOutOfMemory) *oom = true; \
            rc = BP_EHIP;                   \
        }                                   \
    }
    GCHK(hipMalloc((void**)&d_be, (size_t)np * 64));
    GCHK(hipMalloc((void**)&g->d_pts, (size_t)np * sizeof(affine_mem)));
    GCHK(hipMalloc((void**)&g->d_table, g->table_bytes));
    GCHK(hipMalloc((void**)&d_bases, nidx * sizeof(xyzz)));
    GCHK(hipMalloc((void**)&d_basesA, nidx * sizeof(affine_mem)));
    size_t rows_per = BP_GENS_TMP_BYTES / ((size_t)NB * sizeof(xyzz));
    if (rows_per < 1) rows_per = 1;
    if (rows_per > nidx) rows_per = nidx;
    GCHK(hipMalloc((void**)&d_tmp, rows_per * NB * sizeof(xyzz)));
    GCHK(hipMemsetAsync(ctx->d_err, 0, sizeof(int), st));
    GCHK(hipMemcpyAsync(d_be, host.data(), (size_t)np * 64, hipMemcpyHostToDevice, st));
    GCHK(launch_ingest_points(d_be, g->d_pts, np, ctx->d_err, st));
    GCHK(launch_gens_bases(g->d_pts, np, c, W, d_bases, st));
    GCHK(launch_xyzz_to_affine(d_bases, d_basesA, nidx, st));
    for (size_t r0 = 0; r0 < nidx; r0 += rows_per) {  // stream-ordered: a group's scratch is normalized before the next group overwrites it
        const size_t nr = std::min(rows_per, nidx - r0);
        GCHK(launch_gens_multiples(d_basesA + r0, nr, NB, d_tmp, st));
        GCHK(launch_gens_normalize(d_tmp, nr * NB, g->d_table + r0 * NB, st));
    }
    int h_err = 0;
    GCHK(hipMemcpyAsync(&h_err, ctx->d_err, sizeof(int), hipMemcpyDeviceToHost, st));
    GCHK(hipStreamSynchronize(st));
#undef GCHK
    if (rc == BP_OK && h_err) {
        set_error_msg("generator not on curve");
        rc = BP_EPOINT;
    }
    if (d_be) (void)hipFree(d_be);
    if (d_bases) (void)hipFree(d_bases);
    if (d_basesA) (void)hipFree(d_basesA);
    if (d_tmp) (void)hipFree(d_tmp);
    if (rc != BP_OK) {
        bp_gens_destroy(g);
        return rc;
    }
    host.resize((size_t)g->np * 64);
    g->host_be = host;
    *out = g;
    return BP_OK;
}

void bp_gens_destroy(bp_gens* g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    (void)hipDeviceSynchronize();  // asynchronous verifications of any context may still be reading the table
    if (g->d_pts) (void)hipFree(g->d_pts);
    if (g->d_table) (void)hipFree(g->d_table);
    delete g;
}

size_t bp_gens_table_bytes(const bp_gens* g) { return g ? g->table_bytes : 0; }

void bp_verify_set_lanes(bp_ctx* ctx, int lanes) {
    if (ctx) ctx->verify_lanes = lanes;
}
void bp_verify_set_overlap(bp_ctx* ctx, int on) {
    if (!ctx) return;
    CtxLock lk(ctx);
    ctx->verify_overlap = on != 0;
}

// fold the event pairs of the pending call into the accumulators (stream must be idle)
static void prof_collect(bp_ctx* ctx) {
    for (int i = 0; i < ctx->prof_n; i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ctx->prof_ev[i], ctx->prof_ev[i + 1]) == hipSuccess) {
            ctx->prof_ms[ctx->prof_ids[i]] += ms;
            ctx->prof_calls[ctx->prof_ids[i]] += 1;
        }
    }
    ctx->prof_n = 0;
}
static void prof_mark(bp_ctx* ctx, int slot) {
    if (ctx->profiling) (void)hipEventRecord(ctx->prof_ev[slot], ctx->stream);
}

int bp_ctx_wait_event(bp_ctx* ctx, void* hip_event) {
    if (!ctx || !hip_event) return BP_EINVAL;
    CtxLock lk(ctx);
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    hipStream_t side;
    HIPCHK(ctx->side_stream(&side), "hipStreamCreate");
    HIPCHK(hipStreamWaitEvent(ctx->stream, (hipEvent_t)hip_event, 0), "hipStreamWaitEvent");
    HIPCHK(hipStreamWaitEvent(side, (hipEvent_t)hip_event, 0), "hipStreamWaitEvent");
    return BP_OK;
}

int bp_ctx_sync(bp_ctx* ctx) {
    if (!ctx) return BP_EINVAL;
    CtxLock lk(ctx);
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    if (ctx->vside) HIPCHK(hipStreamSynchronize(ctx->vside), "sync");
    ctx->v_side_pending = false;
    ctx->v_calls = 0;  // the next call is one call alone again: on the context's stream, its table kernel on the side stream
    HIPCHK(hipStreamSynchronize(ctx->stream), "sync");
    if (ctx->profiling) prof_collect(ctx);
    if (ctx->async_point_check) {  // an off-curve input point of bp_msm_partial_device_out
        int h = 0;
        HIPCHK(hipMemcpy(&h, ctx->d_err + 1, sizeof(int), hipMemcpyDeviceToHost), "memcpy err");
        ctx->async_point_check = false;
        (void)hipMemsetAsync(ctx->d_err + 1, 0, sizeof(int), ctx->stream);
        if (h) {
            set_error_msg("input point not on curve (bp_msm_partial_device_out)");
            return BP_EPOINT;
        }
    }
    return BP_OK;
}

void bp_profile_enable(bp_ctx* ctx, int on) {
    if (!ctx) return;
    CtxLock lk(ctx);
    (void)hipSetDevice(ctx->device);
    if (on && !ctx->prof_ev[0]) {
        for (int i = 0; i <= bp_ctx::kMaxProf; i++) (void)hipEventCreate(&ctx->prof_ev[i]);
    }
    ctx->profiling = on != 0;
    ctx->prof_n = 0;
}
void bp_profile_reset(bp_ctx* ctx) {
    if (!ctx) return;
    CtxLock lk(ctx);
    for (int i = 0; i < bp_ctx::kMaxProf; i++) {
        ctx->prof_ms[i] = 0;
        ctx->prof_calls[i] = 0;
    }
    ctx->prof_n = 0;
}
int bp_profile_get(bp_ctx* ctx, int kernel_id, double* total_ms, long* launches) {
    if (!ctx || kernel_id < 0 || kernel_id >= bp_ctx::kMaxProf) return BP_EINVAL;
    CtxLock lk(ctx);
    if (total_ms) *total_ms = ctx->prof_ms[kernel_id];
    if (launches) *launches = ctx->prof_calls[kernel_id];
    return BP_OK;
}

// argument checks shared by the batch verifiers: BP_EINVAL where the reference asserts (range_proof.rs:556-562)
// (sizes only: the host-pointer entry points run it BEFORE they read the caller's buffers with these sizes)
static int check_batch_sizes(bp_ctx* ctx, const bp_gens* gens, int n_bits, int m, size_t proof_len, int* k_out, int wip) {
    if (!ctx || !gens || n_bits <= 0 || n_bits > 128 || m <= 0) return BP_EINVAL;
    size_t nm = (size_t)n_bits * m;
    if (nm != gens->nm || (nm & (nm - 1)) != 0 || gens->device != ctx->device) {
        set_error_msg("n_bits*m must be a power of two and equal the generators' nm (range_proof.rs:556-558)");
        return BP_EINVAL;
    }
    int k = 0;
    while (((size_t)1 << k) < nm) k++;
    if (proof_len != (size_t)(wip ? 195 : 292) + 66 * (size_t)k || k > 16) {
        set_error_msg(wip ? "proof_len != 195 + 66*log2(nm)" : "proof_len != 292 + 66*log2(nm)");
        return BP_EINVAL;
    }
    *k_out = k;
    return BP_OK;
}
static int check_batch_args(bp_ctx* ctx, const bp_gens* gens, int n_bits, int m, const void* d_proofs, size_t proof_len, const void* d_coms_xy,
                            size_t nproofs, void* d_verdicts, int* k_out, int wip = 0) {
    int rc = check_batch_sizes(ctx, gens, n_bits, m, proof_len, k_out, wip);
    if (rc != BP_OK) return rc;
    if (nproofs && (!d_proofs || !d_coms_xy || !d_verdicts)) return BP_EINVAL;
    return BP_OK;
}

// Workspaces of one batch verification (device memory beyond the inputs and outputs): the grow-only buffers of the context
// for the ordinary entry points, slices of the file-scope `scratch` array for the no-allocation entry point.
struct VerifyWs {
    affine_mem* pts;   // decompressed proof points + commitments
    int* bad;          // per-proof malformed flags
    tdig_t* dig;       // window digits of the 2nm+2 generator scalars
    fq* var;           // scalars of the proof-specific points
    fq* chal;          // challenges, then the per-proof constants (cst)
    xyzz* fixres;      // y*G, x_u*G (projective), then their affine forms
    xyzz* vartab;      // lane-private window tables of k_bpv_msm
    xyzz* fixpart;     // per-proof sum of the fixed-generator terms (k_bpv_fix -> k_bpv_msm)
};
struct VerifySizes {
    size_t pts, bad, dig, var, chal, fixres, vartab, fixpart;
    int tp;
    size_t total() const {  // as laid out by carve(): every slice starts on a 256-byte boundary
        size_t t = 0;
        for (size_t b : {pts, bad, dig, var, chal, fixres, vartab, fixpart}) t += (b + 255) / 256 * 256;
        return t;
    }
};
static VerifySizes verify_ws_sizes(const bp_ctx* ctx, const bp_gens* gens, int n_bits, int m, int k, size_t nproofs, int wip, bool allow_slice = true) {
    const int NPT = (wip ? 3 : 4) + 2 * k + m, NV = (wip ? 2 : 3) + 2 * k + m;
    VerifySizes z;
    z.pts = nproofs * NPT * sizeof(affine_mem);
    z.bad = nproofs * sizeof(int);
    z.dig = nproofs * (size_t)gens->np * gens->W * sizeof(tdig_t);
    z.var = nproofs * NV * sizeof(fq);
    z.chal = nproofs * (bpv_chal_count(k) + bpv_cst_count(k, m)) * sizeof(fq);
    z.fixres = nproofs * 2 * (sizeof(xyzz) + sizeof(affine_mem));
    // lanes per proof: fewer lanes share more doubling chains (less work) but need more proofs in
    // flight to fill 256 CUs x 4 SIMDs x 2 waves
    // Batches too small to give every SIMD even one wave that way are latency-bound: four lanes per proof-specific point
    // (k_bpv_msm_slice, tp < 0 = minus its lanes per proof; measured on n=64 m=1: 512 proofs 1.21 -> 0.93 ms, 1024 proofs
    // 1.20 -> 0.95 ms, 2048 proofs 1.34 -> 1.55 ms, hence the 65536-lane limit).  bp_verify_set_lanes(ctx, -4) forces that
    // kernel; allow_slice = false keeps the smaller workspace of the ladder kernel (no-allocation entry point).
    // Eight lanes per point (four window ranges per GLV half) while the batch leaves three quarters of the SIMDs without a wave:
    // 36 of the ladder's 64 doublings move into the scalar-independent window tables, which are built beside the scalar chain
    // (k_bpv_chain).  Measured on n=64 m=1, one call alone / two in flight (scripts/experiments/slice8_test.py): 128 proofs
    // 0.82 / 0.41 -> 0.69 / 0.39 ms, 256 proofs 0.79 / 0.41 -> 0.65 / 0.53, 512 proofs 0.82 / 0.65 -> 1.13 / 0.86 (every SIMD
    // already has its wave at four lanes per point), hence the 16384-lane limit.  bp_verify_set_lanes(ctx, -8) forces it.
    int tp = ctx->verify_lanes;
    const int slice_lpp = bpv_slice_lanes(NV), slice_lpp8 = bpv_slice_lanes(NV, 4);
    if (allow_slice && slice_lpp8 <= 128 && (tp == -8 || (tp == 0 && nproofs * (size_t)slice_lpp8 <= 16384))) {
        tp = -slice_lpp8;
    } else if (allow_slice && (tp == -4 || tp == -8 || (tp == 0 && nproofs * (size_t)slice_lpp <= 65536))) {
        tp = -slice_lpp;
    } else if (tp != 2 && tp != 4 && tp != 8 && tp != 16 && tp != 32 && tp != 64) {
        // (re-tuned with two calls in flight and the one-call-alone helper stream: profiles/r02_small_batch.txt E)
        tp = nproofs >= 40960 ? 2 : (nproofs >= 14336 ? 4 : (nproofs >= 6144 ? 8 : 16));
        // aggregated proofs (n*m >= 512): tens of proof-specific points per proof, so take enough lanes for one wave per SIMD
        // (65536 lanes) even with few proofs in the batch; the table kernel picks its own lanes (bpv_fix_lanes).  4096 n=64 m=16
        // proofs: 316 K proofs/s at 32 lanes (the earlier 131072-lane rule), 332 K at 16, 318 K at 8, 305 K at 4.
        if ((size_t)n_bits * m >= 512)
            while (tp < 64 && nproofs * (size_t)tp < 65536) tp *= 2;
    }
    z.tp = tp;
    z.vartab = bpv_msm_vartab_bytes(tp, nproofs);
    z.fixpart = bpv_msm_fixpart_bytes(nproofs);
    return z;
}

static int verify_batch_launch(bp_ctx* ctx, const bp_gens* gens, int n_bits, int m, int k, const void* d_proofs, size_t proof_len,
                               const void* d_coms_xy, size_t nproofs, void* d_verdicts, int wip, const VerifyWs& ws, int tp, hipStream_t st,
                               bool use_helper = false);
static int verify_batch_device_nolock(bp_ctx* ctx, const bp_gens* gens, int n_bits, int m, int k, const void* d_proofs, size_t proof_len,
                                      const void* d_coms_xy, size_t nproofs, void* d_verdicts, int wip = 0, bool overlap = false);

int bp_rangeproof_verify_batch_device(bp_ctx* ctx, const bp_gens* gens, int n_bits, int m, const void* d_proofs, size_t proof_len,
                                      const void* d_coms_xy, size_t nproofs, void* d_verdicts) {
    int k = 0;
    int rc = check_batch_args(ctx, gens, n_bits, m, d_proofs, proof_len, d_coms_xy, nproofs, d_verdicts, &k);
    if (rc != BP_OK || nproofs == 0) return rc;
    std::lock_guard<std::mutex> lk(ctx->mu);  // no CtxLock: this call may run beside the side lane's
    return verify_batch_device_nolock(ctx, gens, n_bits, m, k, d_proofs, proof_len, d_coms_xy, nproofs, d_verdicts, 0, true);
}

// the side lane's stream and event, created at first use
static int verify_overlap_init(bp_ctx* ctx) {
    if (ctx->v_ev_side) return BP_OK;
    hipStream_t s;
    HIPCHK(ctx->side_stream(&s), "hipStreamCreate");
    HIPCHK(hipEventCreateWithFlags(&ctx->v_ev_side, hipEventDisableTiming), "hipEventCreate");
    HIPCHK(hipEventCreateWithFlags(&ctx->v_ev_fork, hipEventDisableTiming), "hipEventCreate");
    HIPCHK(hipEventCreateWithFlags(&ctx->v_ev_join, hipEventDisableTiming), "hipEventCreate");
    return BP_OK;
}

// wip = 1: Bulletproofs+ (RangeProofWIP::aggregated_verify, range_proof_wip.rs:388-609): own decompression order and
// scalar pipeline, the same MSM kernel (its variable terms and comparand follow the point order, verify_kernels.hip)
// overlap = true (the asynchronous device entry points only): every second call goes to the side lane - ctx->vside and the
// workspaces ctx->vws, which nothing else uses - and runs beside the call before and after it on ctx->stream.  Measured on
// 65 536 n=64 m=1 proofs per call: 16.1 -> 14.7 ms per call (the scalar kernels are latency-bound at one wave per SIMD, the
// table kernel leaves a quarter of its workgroups to a second, thin round, the ladder kernel waits on its window tables:
// another batch's waves take those slots).  Not while per-stage timing is on (stages of two calls would interleave), not
// for callers whose inputs are written on ctx->stream (the side lane does not wait for that stream).
static int verify_batch_device_nolock(bp_ctx* ctx, const bp_gens* gens, int n_bits, int m, int k, const void* d_proofs, size_t proof_len,
                                      const void* d_coms_xy, size_t nproofs, void* d_verdicts, int wip, bool overlap) {
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    const VerifySizes z = verify_ws_sizes(ctx, gens, n_bits, m, k, nproofs, wip);
    int rc;
    bool side = false;
    if (overlap && ctx->verify_overlap && !ctx->profiling) {
        if ((rc = verify_overlap_init(ctx)) != BP_OK) return rc;
        side = (ctx->v_calls++ & 1u) != 0;
    } else {
        ctx->join_side();  // a serial call after overlapped ones (overlap switched off, per-stage timing switched on)
    }
    // One call ALONE (nothing pending on the side lane: a synchronous caller, or the first call of a stream of calls): its
    // fixed-generator kernel goes to the side stream, beside the proof-specific kernel (launch_bpv_msm).  Not under per-stage
    // timing, whose figures are the kernels with the GPU to themselves.
    bool use_helper = false;
    if (!side && !ctx->v_side_pending && ctx->verify_overlap && !ctx->profiling) {
        if ((rc = verify_overlap_init(ctx)) != BP_OK) return rc;
        use_helper = true;
    }
    DevBuf* w = side ? ctx->vws : nullptr;
    DevBuf &b_pts = w ? w[0] : ctx->scratch[0], &b_bad = w ? w[1] : ctx->scratch[1], &b_dig = w ? w[2] : ctx->scratch[2],
           &b_var = w ? w[3] : ctx->scratch[3], &b_s = w ? w[4] : ctx->scratch[4], &b_fixres = w ? w[5] : ctx->fixres,
           &b_vartab = w ? w[6] : ctx->vartab, &b_fixpart = w ? w[7] : ctx->fixpart;
    if ((rc = b_pts.reserve(z.pts)) != BP_OK) return rc;
    if ((rc = b_bad.reserve(z.bad)) != BP_OK) return rc;
    if ((rc = b_dig.reserve(z.dig)) != BP_OK) return rc;
    if ((rc = b_var.reserve(z.var)) != BP_OK) return rc;
    if ((rc = b_s.reserve(z.chal)) != BP_OK) return rc;
    if ((rc = b_fixres.reserve(z.fixres)) != BP_OK) return rc;
    if ((rc = b_vartab.reserve(z.vartab)) != BP_OK) return rc;
    if ((rc = b_fixpart.reserve(z.fixpart)) != BP_OK) return rc;
    VerifyWs ws{(affine_mem*)b_pts.p, (int*)b_bad.p, (tdig_t*)b_dig.p, (fq*)b_var.p, (fq*)b_s.p, (xyzz*)b_fixres.p, (xyzz*)b_vartab.p,
                (xyzz*)b_fixpart.p};
    rc = verify_batch_launch(ctx, gens, n_bits, m, k, d_proofs, proof_len, d_coms_xy, nproofs, d_verdicts, wip, ws, z.tp, side ? ctx->vside : ctx->stream,
                             use_helper);
    if (side) {
        HIPCHK(hipEventRecord(ctx->v_ev_side, ctx->vside), "record event");
        ctx->v_side_pending = true;
    }
    return rc;
}

// the kernels of one batch verification on stream st (the context's, or the side lane's); allocates nothing
static int verify_batch_launch(bp_ctx* ctx, const bp_gens* gens, int n_bits, int m, int k, const void* d_proofs, size_t proof_len,
                               const void* d_coms_xy, size_t nproofs, void* d_verdicts, int wip, const VerifyWs& ws, int tp, hipStream_t st,
                               bool use_helper) {
    fq* d_chal = ws.chal;
    fq* d_cst = d_chal + nproofs * bpv_chal_count(k);
    xyzz* d_fixres = ws.fixres;
    affine_mem* d_fixaff = (affine_mem*)(d_fixres + nproofs * 2);
    HIPCHK(hipMemsetAsync(ws.bad, 0, nproofs * sizeof(int), st), "memset");
    if (ctx->profiling && ctx->prof_n) {  // previous call not collected yet: drain it first
        HIPCHK(hipStreamSynchronize(st), "sync");
        prof_collect(ctx);
    }
    prof_mark(ctx, 0);
    if (wip)
        HIPCHK(launch_wipv_decompress((const uint8_t*)d_proofs, proof_len, (const uint8_t*)d_coms_xy, k, m, nproofs, ws.pts, ws.bad, st),
               "wipv_decompress");
    else
        HIPCHK(launch_bpv_decompress((const uint8_t*)d_proofs, proof_len, (const uint8_t*)d_coms_xy, k, m, nproofs, ws.pts, ws.bad, st),
               "bpv_decompress");
    prof_mark(ctx, 1);
    // small batches: the ladders' scalar-independent window tables are built in the constants kernel's launch (into vartab)
    const int lpp = bpv_small_fused(tp), nparts = bpv_small_parts(tp, (wip ? 2 : 3) + 2 * k + m);
    if (wip)
        HIPCHK(launch_wipv_scalar_pipeline((const uint8_t*)d_proofs, proof_len, n_bits, m, k, nproofs, ws.pts, gens->d_pts, gens->d_table, gens->c,
                                           gens->W, ws.dig, nullptr, ws.var, d_chal, d_cst, d_fixres, d_fixaff, ws.bad, st, gens->ibase, (int)gens->nm,
                                           lpp ? ws.vartab : nullptr, lpp, nparts ? nparts : 2),
               "wipv_scalars");
    else
        HIPCHK(launch_bpv_scalar_pipeline((const uint8_t*)d_proofs, proof_len, n_bits, m, k, nproofs, ws.pts, gens->d_pts, gens->d_table, gens->c,
                                          gens->W, ws.dig, nullptr, ws.var, d_chal, d_cst, d_fixres, d_fixaff, ws.bad, st, gens->ibase, (int)gens->nm,
                                          lpp ? ws.vartab : nullptr, lpp, nparts ? nparts : 2),
               "bpv_scalars");
    prof_mark(ctx, 2);
    HIPCHK(launch_bpv_msm(tp, n_bits, m, k, nproofs, ws.pts, gens->d_table, gens->c, gens->W, ws.dig, ws.var, ws.vartab, ws.fixpart, ws.bad,
                          (uint8_t*)d_verdicts, st, wip, (int)gens->nm, nparts, use_helper ? ctx->vside : nullptr, ctx->v_ev_fork,
                          ctx->v_ev_join),
           "bpv_msm");
    prof_mark(ctx, 3);
    if (ctx->profiling) {
        ctx->prof_ids[0] = BP_PROF_BPV_DECOMPRESS;
        ctx->prof_ids[1] = BP_PROF_BPV_SCALARS;
        ctx->prof_ids[2] = BP_PROF_BPV_MSM;
        ctx->prof_n = 3;
    }
    return BP_OK;
}

// ---- the no-allocation entry point ---------------------------------------------------------------
// Callers that cannot call hipMalloc (HIP graph capture, tight serving loops, harnesses that forbid it) get the
// batch verifier with its workspaces in a file-scope device array instead of the context's grow-only buffers.
// Every GPU the library runs on holds BP_NOALLOC_SCRATCH_BYTES of it until the process ends.  The array's name is
// only a host-side stand-in: the per-device address comes from hipGetSymbolAddress (noalloc_init_device, called by
// bp_ctx_create with the device current) and only that pointer is ever handed to kernels / memsets.
static __device__ __align__(256) unsigned char scratch[BP_NOALLOC_SCRATCH_BYTES];
static const int kMaxDevices = 64;
static unsigned char* g_scratch_ptr[kMaxDevices];
static hipEvent_t g_scratch_ev[kMaxDevices];   // recorded behind the last no-allocation call of the device: the next call's stream waits for it
static bool g_scratch_busy[kMaxDevices];       // an event has been recorded
static std::mutex g_scratch_mu;

}  // extern "C"
namespace bp {
int noalloc_init_device(int device) {  // the device must be current
    if (device < 0 || device >= kMaxDevices) return BP_OK;  // such a device simply has no no-allocation entry point
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (g_scratch_ptr[device]) return BP_OK;
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) {
        // an optional feature must not take every other entry point down with it: the pointer stays null and only
        // bp_rangeproof_verify_batch_device_noalloc reports BP_EHIP on this device
        (void)hipGetLastError();
        return BP_OK;
    }
    if (hipEventCreateWithFlags(&g_scratch_ev[device], hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        return BP_OK;
    }
    g_scratch_ptr[device] = (unsigned char*)p;
    return BP_OK;
}
}  // namespace bp
extern "C" {

size_t bp_noalloc_scratch_bytes(void) { return BP_NOALLOC_SCRATCH_BYTES; }

int bp_rangeproof_verify_batch_device_noalloc(bp_ctx* ctx, const bp_gens* gens, int n_bits, int m, const void* d_proofs, size_t proof_len,
                                              const void* d_coms_xy, size_t nproofs, void* d_verdicts) {
    int k = 0;
    int rc = check_batch_args(ctx, gens, n_bits, m, d_proofs, proof_len, d_coms_xy, nproofs, d_verdicts, &k);
    if (rc != BP_OK || nproofs == 0) return rc;
    CtxLock lk(ctx);
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    unsigned char* base = ctx->device < kMaxDevices ? g_scratch_ptr[ctx->device] : nullptr;
    if (!base) {
        set_error_msg("no-allocation entry point: this device's scratch address was not resolved at bp_ctx_create");
        return BP_EHIP;
    }
    VerifySizes z = verify_ws_sizes(ctx, gens, n_bits, m, k, nproofs, 0);
    // the small-batch kernel's window tables take twice the workspace per proof: fall back to the ladder kernel before refusing
    if (z.tp < 0 && z.total() > (size_t)BP_NOALLOC_SCRATCH_BYTES) z = verify_ws_sizes(ctx, gens, n_bits, m, k, nproofs, 0, false);
    if (nproofs > ((size_t)1 << 24) || z.total() > (size_t)BP_NOALLOC_SCRATCH_BYTES) {
        char msg[200];
        snprintf(msg, sizeof msg, "no-allocation entry point: the batch needs %zu bytes of workspace, the fixed scratch holds %zu; split the batch",
                 z.total(), (size_t)BP_NOALLOC_SCRATCH_BYTES);
        set_error_msg(msg);
        return BP_EINVAL;
    }
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        unsigned char* p = base + off;
        off += (bytes + 255) / 256 * 256;
        return p;
    };
    VerifyWs ws;
    ws.pts = (affine_mem*)carve(z.pts);
    ws.bad = (int*)carve(z.bad);
    ws.dig = (tdig_t*)carve(z.dig);
    ws.var = (fq*)carve(z.var);
    ws.chal = (fq*)carve(z.chal);
    ws.fixres = (xyzz*)carve(z.fixres);
    ws.vartab = (xyzz*)carve(z.vartab);
    ws.fixpart = (xyzz*)carve(z.fixpart);
    // The array keeps the previous call's contents.  Only `bad` must start at zero (verify_batch_launch clears it on the
    // stream); every other slice is written by the kernels of this call before it is read.
    // Calls of OTHER contexts / threads on this GPU use the same array: order this call's stream after the last one's
    // kernels (one event per device, waited for and re-recorded under the device's mutex), so overlapping calls serialise.
    std::lock_guard<std::mutex> glk(g_scratch_mu);
    if (g_scratch_busy[ctx->device]) HIPCHK(hipStreamWaitEvent(ctx->stream, g_scratch_ev[ctx->device], 0), "hipStreamWaitEvent");
    rc = verify_batch_launch(ctx, gens, n_bits, m, k, d_proofs, proof_len, d_coms_xy, nproofs, d_verdicts, 0, ws, z.tp, ctx->stream);
    HIPCHK(hipEventRecord(g_scratch_ev[ctx->device], ctx->stream), "hipEventRecord");
    g_scratch_busy[ctx->device] = true;
    return rc;
}

int bp_wip_rangeproof_verify_batch_device(bp_ctx* ctx, const bp_gens* gens, int n_bits, int m, const void* d_proofs, size_t proof_len,
                                          const void* d_coms_xy, size_t nproofs, void* d_verdicts) {
    int k = 0;
    int rc = check_batch_args(ctx, gens, n_bits, m, d_proofs, proof_len, d_coms_xy, nproofs, d_verdicts, &k, 1);
    if (rc != BP_OK || nproofs == 0) return rc;
    std::lock_guard<std::mutex> lk(ctx->mu);  // no CtxLock: this call may run beside the side lane's
    return verify_batch_device_nolock(ctx, gens, n_bits, m, k, d_proofs, proof_len, d_coms_xy, nproofs, d_verdicts, 1, true);
}

// Random-linear-combination batch verification (rlc_kernels.hip).  Synchronous: returns with the verdicts
// written.  *used_fallback (nullable) = 1 when the combined check failed and the per-proof kernel decided.
static int verify_batch_rlc_device_impl(bp_ctx* ctx, const bp_gens* gens, int n_bits, int m, const void* d_proofs, size_t proof_len,
                                        const void* d_coms_xy, size_t nproofs, const uint8_t seed[32], void* d_verdicts, int* used_fallback,
                                        int wip) {
    int k = 0;
    int rc = check_batch_args(ctx, gens, n_bits, m, d_proofs, proof_len, d_coms_xy, nproofs, d_verdicts, &k, wip);
    if (rc != BP_OK) return rc;
    if (!seed) return BP_EINVAL;
    if (used_fallback) *used_fallback = 0;
    if (nproofs == 0) return BP_OK;
    CtxLock lk(ctx);
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    hipStream_t st = ctx->stream;
    const int NPT = (wip ? 3 : 4) + 2 * k + m, NV = (wip ? 2 : 3) + 2 * k + m, np = gens->np, W = gens->W;
    if (nproofs * (size_t)NPT >= ((size_t)1 << 30)) {
        set_error_msg("rlc batch too large for one MSM");
        return BP_EINVAL;
    }
    DevBuf &b_pts = ctx->scratch[0], &b_bad = ctx->scratch[1], &b_scF = ctx->scratch[2], &b_var = ctx->scratch[3], &b_s = ctx->scratch[4];
    DevBuf &b_rho = ctx->rlc[0], &b_scbe = ctx->rlc[1], &b_misc = ctx->rlc[2];
    const size_t nchunks = rlc_colsum_chunks(nproofs);
    if ((rc = b_pts.reserve(nproofs * NPT * sizeof(affine_mem))) != BP_OK) return rc;
    if ((rc = b_bad.reserve(nproofs * sizeof(int))) != BP_OK) return rc;
    if ((rc = b_scF.reserve(nproofs * (size_t)np * sizeof(fqm))) != BP_OK) return rc;
    if ((rc = b_var.reserve(nproofs * NV * sizeof(fq))) != BP_OK) return rc;
    if ((rc = b_s.reserve(nproofs * (bpv_chal_count(k) + bpv_cst_count(k, m)) * sizeof(fq))) != BP_OK) return rc;
    if ((rc = ctx->fixres.reserve(nproofs * 2 * (sizeof(xyzz) + sizeof(affine_mem)))) != BP_OK) return rc;
    if ((rc = b_rho.reserve(nproofs * sizeof(fqm))) != BP_OK) return rc;
    if ((rc = b_scbe.reserve(nproofs * (size_t)NPT * 32)) != BP_OK) return rc;
    // misc: chunk sums | fixed scalars (BE) | seed | generator index 0..np-1 | term offsets {0, np} | extras offsets {0, 0} |
    //       fixed-part result (BE) | its limb form | all_ok
    const size_t o_part = 0, o_fbe = o_part + (size_t)np * nchunks * sizeof(fqm), o_seed = o_fbe + (size_t)np * 32, o_gidx = o_seed + 32,
                 o_toff = o_gidx + (size_t)np * 4, o_xoff = o_toff + 8, o_res = (o_xoff + 8 + 63) / 64 * 64, o_init = o_res + 64,
                 o_ok = o_init + sizeof(affine_mem), misc_bytes = o_ok + 64;
    if ((rc = b_misc.reserve(misc_bytes)) != BP_OK) return rc;
    if ((rc = ctx->fb[1].reserve((size_t)np * W * sizeof(tdig_t))) != BP_OK) return rc;
    int nblk = (int)(((size_t)np * W + 255) / 256);
    if (nblk > 128) nblk = 128;
    if ((rc = ctx->fb[2].reserve((size_t)nblk * sizeof(xyzz))) != BP_OK) return rc;
    fq* d_chal = (fq*)b_s.p;
    fq* d_cst = d_chal + nproofs * bpv_chal_count(k);
    xyzz* d_fixres = (xyzz*)ctx->fixres.p;
    affine_mem* d_fixaff = (affine_mem*)(d_fixres + nproofs * 2);
    uint8_t* misc = (uint8_t*)b_misc.p;
    std::vector<uint32_t> h_idx((size_t)np + 4);
    for (int i = 0; i < np; i++) h_idx[i] = (uint32_t)i;
    h_idx[np] = 0;
    h_idx[np + 1] = (uint32_t)np;
    h_idx[np + 2] = 0;
    h_idx[np + 3] = 0;
    HIPCHK(hipMemcpyAsync(misc + o_seed, seed, 32, hipMemcpyHostToDevice, st), "memcpy seed");
    HIPCHK(hipMemcpyAsync(misc + o_gidx, h_idx.data(), ((size_t)np + 4) * 4, hipMemcpyHostToDevice, st), "memcpy index");
    HIPCHK(hipMemsetAsync(b_bad.p, 0, nproofs * sizeof(int), st), "memset");
    HIPCHK(hipMemsetAsync(ctx->d_err, 0, sizeof(int), st), "memset");
    if (wip) {
        HIPCHK(launch_wipv_decompress((const uint8_t*)d_proofs, proof_len, (const uint8_t*)d_coms_xy, k, m, nproofs, (affine_mem*)b_pts.p,
                                      (int*)b_bad.p, st),
               "wipv_decompress");
        HIPCHK(launch_wipv_scalar_pipeline((const uint8_t*)d_proofs, proof_len, n_bits, m, k, nproofs, (const affine_mem*)b_pts.p, gens->d_pts,
                                           gens->d_table, gens->c, W, nullptr, (fqm*)b_scF.p, (fq*)b_var.p, d_chal, d_cst, d_fixres, d_fixaff,
                                           (int*)b_bad.p, st, gens->ibase, (int)gens->nm),
               "wipv_scalars");
    } else {
        HIPCHK(launch_bpv_decompress((const uint8_t*)d_proofs, proof_len, (const uint8_t*)d_coms_xy, k, m, nproofs, (affine_mem*)b_pts.p,
                                     (int*)b_bad.p, st),
               "bpv_decompress");
        HIPCHK(launch_bpv_scalar_pipeline((const uint8_t*)d_proofs, proof_len, n_bits, m, k, nproofs, (const affine_mem*)b_pts.p, gens->d_pts,
                                          gens->d_table, gens->c, W, nullptr, (fqm*)b_scF.p, (fq*)b_var.p, d_chal, d_cst, d_fixres, d_fixaff,
                                          (int*)b_bad.p, st, gens->ibase, (int)gens->nm),
               "bpv_scalars");
    }
    HIPCHK(launch_rlc_weights(misc + o_seed, nproofs, (const int*)b_bad.p, (fqm*)b_rho.p, st), "rlc_weights");
    HIPCHK(launch_rlc_var_scalars(k, m, nproofs, (const fqm*)b_rho.p, (const fq*)b_var.p, (uint8_t*)b_scbe.p, st, wip), "rlc_var_scalars");
    HIPCHK(launch_rlc_colsum(np, nproofs, (const fqm*)b_rho.p, (const fqm*)b_scF.p, (const int*)b_bad.p, (fqm*)(misc + o_part), misc + o_fbe, st),
           "rlc_colsum");
    // the 2nm+2 generator terms: one fixed-base MSM from the table; its value is the init of the big MSM
    HIPCHK(launch_mb_digits(misc + o_fbe, (size_t)np, gens->c, W, (tdig_t*)ctx->fb[1].p, st), "digits");
    HIPCHK(launch_fb_msm(1, gens->d_table, W, gens->NB, (const tdig_t*)ctx->fb[1].p, (const uint32_t*)(misc + o_gidx),
                         (const uint32_t*)(misc + o_toff), nullptr, nullptr, (const uint32_t*)(misc + o_xoff), nblk, 0, nullptr,
                         (xyzz*)ctx->fb[2].p, misc + o_res, st),
           "fb_msm");
    HIPCHK(launch_ingest_points(misc + o_res, (affine_mem*)(misc + o_init), 1, ctx->d_err, st), "ingest");
    if ((rc = msm_on_device(ctx, (const affine_mem*)b_pts.p, (const uint8_t*)b_scbe.p, nproofs * (size_t)NPT, (const affine_mem*)(misc + o_init))) != BP_OK)
        return rc;
    HIPCHK(launch_rlc_verdicts(nproofs, (const uint8_t*)ctx->result.p + sizeof(affine_mem), (const int*)b_bad.p, (uint8_t*)d_verdicts,
                               (int*)(misc + o_ok), st),
           "rlc_verdicts");
    int all_ok = 0;
    HIPCHK(hipMemcpyAsync(&all_ok, misc + o_ok, sizeof(int), hipMemcpyDeviceToHost, st), "memcpy flag");
    HIPCHK(hipStreamSynchronize(st), "sync");
    if (all_ok) return BP_OK;
    // some well-formed proof is invalid: the per-proof kernel decides (identical verdicts by construction)
    if (used_fallback) *used_fallback = 1;
    if ((rc = verify_batch_device_nolock(ctx, gens, n_bits, m, k, d_proofs, proof_len, d_coms_xy, nproofs, d_verdicts, wip)) != BP_OK) return rc;
    HIPCHK(hipStreamSynchronize(st), "sync");
    return BP_OK;
}

int bp_rangeproof_verify_batch_rlc_device(bp_ctx* ctx, const bp_gens* gens, int n_bits, int m, const void* d_proofs, size_t proof_len,
                                          const void* d_coms_xy, size_t nproofs, const uint8_t seed[32], void* d_verdicts, int* used_fallback) {
    return verify_batch_rlc_device_impl(ctx, gens, n_bits, m, d_proofs, proof_len, d_coms_xy, nproofs, seed, d_verdicts, used_fallback, 0);
}
int bp_wip_rangeproof_verify_batch_rlc_device(bp_ctx* ctx, const bp_gens* gens, int n_bits, int m, const void* d_proofs, size_t proof_len,
                                              const void* d_coms_xy, size_t nproofs, const uint8_t seed[32], void* d_verdicts,
                                              int* used_fallback) {
    return verify_batch_rlc_device_impl(ctx, gens, n_bits, m, d_proofs, proof_len, d_coms_xy, nproofs, seed, d_verdicts, used_fallback, 1);
}

int bp_msm_batch(bp_ctx* ctx, const bp_gens* gens, const bp_msm_job* jobs, size_t njobs, uint8_t* out_xy) {
    if (!ctx || !gens || (njobs && (!jobs || !out_xy))) return BP_EINVAL;
    if (njobs == 0) return BP_OK;
    if (gens->device != ctx->device) return BP_EINVAL;
    const int np = gens->np;
    // pack on the host: generator scalars [njobs][np][32] (zero rows for jobs without), extras concatenated
    std::vector<uint8_t> gsc(njobs * (size_t)np * 32, 0), has(njobs, 0);
    std::vector<uint32_t> xoff(njobs + 1, 0);
    size_t nx = 0;
    for (size_t j = 0; j < njobs; j++) {
        if (jobs[j].n_extra && (!jobs[j].extra_points_xy || !jobs[j].extra_scalars_be)) return BP_EINVAL;
        nx += jobs[j].n_extra;
        if (nx >= ((size_t)1 << 31)) return BP_EINVAL;
        xoff[j + 1] = (uint32_t)nx;
        if (jobs[j].gen_scalars_be) {
            memcpy(&gsc[j * (size_t)np * 32], jobs[j].gen_scalars_be, (size_t)np * 32);
            has[j] = 1;
        }
    }
    std::vector<uint8_t> xp(nx ? nx * 64 : 64), xs(nx ? nx * 32 : 32);
    for (size_t j = 0; j < njobs; j++) {
        if (!jobs[j].n_extra) continue;
        memcpy(&xp[(size_t)xoff[j] * 64], jobs[j].extra_points_xy, jobs[j].n_extra * 64);
        memcpy(&xs[(size_t)xoff[j] * 32], jobs[j].extra_scalars_be, jobs[j].n_extra * 32);
    }
    CtxLock lk(ctx);
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    hipStream_t st = ctx->stream;
    int rc;
    DevBuf &b_gsc = ctx->scratch[0], &b_dig = ctx->scratch[2], &b_x = ctx->scratch[3], &b_misc = ctx->scratch[4], &b_out = ctx->scratch[5];
    const size_t nxa = nx ? nx : 1;
    if ((rc = b_gsc.reserve(gsc.size())) != BP_OK) return rc;
    if ((rc = b_dig.reserve(njobs * (size_t)np * gens->W * sizeof(tdig_t))) != BP_OK) return rc;
    if ((rc = b_x.reserve(nxa * (64 + 32 + sizeof(affine_mem)))) != BP_OK) return rc;
    if ((rc = b_misc.reserve((njobs + 1) * 4 + njobs + 16)) != BP_OK) return rc;
    if ((rc = b_out.reserve(njobs * (sizeof(xyzz) + sizeof(affine_mem) + 64))) != BP_OK) return rc;
    if ((rc = ctx->vartab.reserve(mb_vartab_bytes(njobs))) != BP_OK) return rc;
    uint8_t* d_xbe = (uint8_t*)b_x.p;
    uint8_t* d_xsc = d_xbe + nxa * 64;
    affine_mem* d_xpts = (affine_mem*)(d_xsc + nxa * 32);
    uint32_t* d_xoff = (uint32_t*)b_misc.p;
    uint8_t* d_has = (uint8_t*)(d_xoff + njobs + 1);
    xyzz* d_res = (xyzz*)b_out.p;
    affine_mem* d_aff = (affine_mem*)(d_res + njobs);
    uint8_t* d_be = (uint8_t*)(d_aff + njobs);
    HIPCHK(hipMemsetAsync(ctx->d_err, 0, sizeof(int), st), "memset err");
    HIPCHK(hipMemcpyAsync(b_gsc.p, gsc.data(), gsc.size(), hipMemcpyHostToDevice, st), "memcpy gen scalars");
    HIPCHK(hipMemcpyAsync(d_xbe, xp.data(), nxa * 64, hipMemcpyHostToDevice, st), "memcpy extra points");
    HIPCHK(hipMemcpyAsync(d_xsc, xs.data(), nxa * 32, hipMemcpyHostToDevice, st), "memcpy extra scalars");
    HIPCHK(hipMemcpyAsync(d_xoff, xoff.data(), (njobs + 1) * 4, hipMemcpyHostToDevice, st), "memcpy offsets");
    HIPCHK(hipMemcpyAsync(d_has, has.data(), njobs, hipMemcpyHostToDevice, st), "memcpy flags");
    HIPCHK(launch_ingest_points(d_xbe, d_xpts, nx, ctx->d_err, st), "ingest");
    HIPCHK(launch_mb_digits((const uint8_t*)b_gsc.p, njobs * (size_t)np, gens->c, gens->W, (tdig_t*)b_dig.p, st), "mb_digits");
    HIPCHK(launch_mb_msm(njobs, np, gens->d_table, gens->c, gens->W, (const tdig_t*)b_dig.p, d_has, d_xpts, d_xsc, d_xoff, (xyzz*)ctx->vartab.p,
                         d_res, st),
           "mb_msm");
    HIPCHK(launch_xyzz_to_affine(d_res, d_aff, njobs, st), "normalize");
    HIPCHK(launch_export_points(d_aff, d_be, njobs, st), "export");
    std::vector<uint8_t> tmp(njobs * 64);
    int h_err = 0;
    HIPCHK(hipMemcpyAsync(tmp.data(), d_be, njobs * 64, hipMemcpyDeviceToHost, st), "memcpy out");
    HIPCHK(hipMemcpyAsync(&h_err, ctx->d_err, sizeof(int), hipMemcpyDeviceToHost, st), "memcpy err");
    HIPCHK(hipStreamSynchronize(st), "sync");
    if (h_err) {
        set_error_msg("input point not on curve");
        return BP_EPOINT;
    }
    memcpy(out_xy, tmp.data(), njobs * 64);
    return BP_OK;
}

// The batch verifier entirely on ctx->stream, for callers whose inputs are produced on that stream (the host-pointer entry
// points stage them with hipMemcpyAsync): the overlapped form's side stream would not wait for them.
static int verify_batch_device_serial(bp_ctx* ctx, const bp_gens* gens, int n_bits, int m, const void* d_proofs, size_t proof_len,
                                      const void* d_coms_xy, size_t nproofs, void* d_verdicts, int wip) {
    int k = 0;
    int rc = check_batch_args(ctx, gens, n_bits, m, d_proofs, proof_len, d_coms_xy, nproofs, d_verdicts, &k, wip);
    if (rc != BP_OK || nproofs == 0) return rc;
    CtxLock lk(ctx);
    return verify_batch_device_nolock(ctx, gens, n_bits, m, k, d_proofs, proof_len, d_coms_xy, nproofs, d_verdicts, wip, false);
}

// host buffers -> device staging (scratch[5..7]); shared by the two host-pointer batch verifiers
static int stage_batch(bp_ctx* ctx, const uint8_t* proofs, size_t proof_len, const uint8_t* coms_xy, int m, size_t nproofs, void** d_p, void** d_c,
                       void** d_v) {
    CtxLock lk(ctx);
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    int rc;
    if ((rc = ctx->scratch[5].reserve(nproofs * proof_len)) != BP_OK) return rc;
    if ((rc = ctx->scratch[6].reserve(nproofs * (size_t)m * 64)) != BP_OK) return rc;
    if ((rc = ctx->scratch[7].reserve(nproofs)) != BP_OK) return rc;
    *d_p = ctx->scratch[5].p;
    *d_c = ctx->scratch[6].p;
    *d_v = ctx->scratch[7].p;
    HIPCHK(hipMemcpyAsync(*d_p, proofs, nproofs * proof_len, hipMemcpyHostToDevice, ctx->stream), "memcpy proofs");
    HIPCHK(hipMemcpyAsync(*d_c, coms_xy, nproofs * (size_t)m * 64, hipMemcpyHostToDevice, ctx->stream), "memcpy coms");
    return BP_OK;
}

static int verify_batch_rlc_host(bp_ctx* ctx, const bp_gens* gens, int n_bits, int m, const uint8_t* proofs, size_t proof_len,
                                 const uint8_t* coms_xy, size_t nproofs, const uint8_t seed[32], uint8_t* verdicts, int* used_fallback, int wip);
int bp_rangeproof_verify_batch_rlc(bp_ctx* ctx, const bp_gens* gens, int n_bits, int m, const uint8_t* proofs, size_t proof_len,
                                   const uint8_t* coms_xy, size_t nproofs, const uint8_t seed[32], uint8_t* verdicts, int* used_fallback) {
    return verify_batch_rlc_host(ctx, gens, n_bits, m, proofs, proof_len, coms_xy, nproofs, seed, verdicts, used_fallback, 0);
}
int bp_wip_rangeproof_verify_batch_rlc(bp_ctx* ctx, const bp_gens* gens, int n_bits, int m, const uint8_t* proofs, size_t proof_len,
                                       const uint8_t* coms_xy, size_t nproofs, const uint8_t seed[32], uint8_t* verdicts, int* used_fallback) {
    return verify_batch_rlc_host(ctx, gens, n_bits, m, proofs, proof_len, coms_xy, nproofs, seed, verdicts, used_fallback, 1);
}
static int verify_batch_rlc_host(bp_ctx* ctx, const bp_gens* gens, int n_bits, int m, const uint8_t* proofs, size_t proof_len,
                                 const uint8_t* coms_xy, size_t nproofs, const uint8_t seed[32], uint8_t* verdicts, int* used_fallback, int wip) {
    if (!seed) return BP_EINVAL;
    int k = 0;
    int rc = check_batch_sizes(ctx, gens, n_bits, m, proof_len, &k, wip);
    if (rc != BP_OK) return rc;
    if (used_fallback) *used_fallback = 0;
    if (nproofs == 0) return BP_OK;
    if (!proofs || !coms_xy || !verdicts) return BP_EINVAL;
    void *d_p, *d_c, *d_v;
    rc = stage_batch(ctx, proofs, proof_len, coms_xy, m, nproofs, &d_p, &d_c, &d_v);
    if (rc != BP_OK) return rc;
    rc = verify_batch_rlc_device_impl(ctx, gens, n_bits, m, d_p, proof_len, d_c, nproofs, seed, d_v, used_fallback, wip);
    if (rc != BP_OK) return rc;
    std::vector<uint8_t> tmp(nproofs);
    HIPCHK(hipMemcpyAsync(tmp.data(), d_v, nproofs, hipMemcpyDeviceToHost, ctx->stream), "memcpy verdicts");
    HIPCHK(hipStreamSynchronize(ctx->stream), "sync");
    memcpy(verdicts, tmp.data(), nproofs);
    return BP_OK;
}

int bp_rangeproof_verify_batch(bp_ctx* ctx, const bp_gens* gens, int n_bits, int m, const uint8_t* proofs, size_t proof_len,
                               const uint8_t* coms_xy, size_t nproofs, uint8_t* verdicts) {
    int k0 = 0;
    if (int rcs = check_batch_sizes(ctx, gens, n_bits, m, proof_len, &k0, 0)) return rcs;
    if (nproofs == 0) return BP_OK;
    if (!proofs || !coms_xy || !verdicts) return BP_EINVAL;
    void *d_p, *d_c, *d_v;
    if (int rcs = stage_batch(ctx, proofs, proof_len, coms_xy, m, nproofs, &d_p, &d_c, &d_v)) return rcs;
    int rc = verify_batch_device_serial(ctx, gens, n_bits, m, d_p, proof_len, d_c, nproofs, d_v, 0);  // staged on ctx->stream: no overlap
    if (rc != BP_OK) return rc;
    std::vector<uint8_t> tmp(nproofs);
    HIPCHK(hipMemcpyAsync(tmp.data(), d_v, nproofs, hipMemcpyDeviceToHost, ctx->stream), "memcpy verdicts");
    HIPCHK(hipStreamSynchronize(ctx->stream), "sync");
    memcpy(verdicts, tmp.data(), nproofs);
    return BP_OK;
}

int bp_wip_rangeproof_verify_batch(bp_ctx* ctx, const bp_gens* gens, int n_bits, int m, const uint8_t* proofs, size_t proof_len,
                                   const uint8_t* coms_xy, size_t nproofs, uint8_t* verdicts) {
    int k0 = 0;
    if (int rcs = check_batch_sizes(ctx, gens, n_bits, m, proof_len, &k0, 1)) return rcs;
    if (nproofs == 0) return BP_OK;
    if (!proofs || !coms_xy || !verdicts) return BP_EINVAL;
    void *d_p, *d_c, *d_v;
    if (int rcs = stage_batch(ctx, proofs, proof_len, coms_xy, m, nproofs, &d_p, &d_c, &d_v)) return rcs;
    int rc = verify_batch_device_serial(ctx, gens, n_bits, m, d_p, proof_len, d_c, nproofs, d_v, 1);
    if (rc != BP_OK) return rc;
    std::vector<uint8_t> tmp(nproofs);
    HIPCHK(hipMemcpyAsync(tmp.data(), d_v, nproofs, hipMemcpyDeviceToHost, ctx->stream), "memcpy verdicts");
    HIPCHK(hipStreamSynchronize(ctx->stream), "sync");
    memcpy(verdicts, tmp.data(), nproofs);
    return BP_OK;
}


// Batches mixing aggregation sizes (proof_off[] + m_per_proof[]): proofs are grouped by m on the host, each
// group is one launch of the batch verifier over the PREFIXES g_vec[..n m_i], h_vec[..n m_i] of the generators (what a
// caller of the reference passes as slices, range_proof.rs:556-557); verdicts return in the caller's order.
static int verify_batch_mixed(bp_ctx* ctx, const bp_gens* gens, int n_bits, const uint8_t* proofs, const size_t* proof_off, const uint8_t* coms_xy,
                              const uint32_t* m_per_proof, size_t nproofs, uint8_t* verdicts, int wip) {
    if (!ctx || !gens || n_bits <= 0 || n_bits > 128 || gens->device != ctx->device) return BP_EINVAL;
    if (nproofs == 0) return BP_OK;
    if (!proofs || !proof_off || !coms_xy || !m_per_proof || !verdicts) return BP_EINVAL;
    struct Group {
        int m, k;
        std::vector<size_t> idx;
    };
    std::vector<Group> groups;
    std::vector<size_t> com_off(nproofs + 1, 0);
    for (size_t i = 0; i < nproofs; i++) {
        const uint32_t m = m_per_proof[i];
        const size_t nm = (size_t)n_bits * m;
        if (m == 0 || nm > gens->nm || (nm & (nm - 1)) != 0) {
            set_error_msg("mixed batch: every n_bits*m must be a power of two not above the generators' nm (range_proof.rs:556-558)");
            return BP_EINVAL;
        }
        int k = 0;
        while (((size_t)1 << k) < nm) k++;
        if (k > 16) {  // the same cap as check_batch_sizes: the constants kernels hold k + 2 prefix scalars per lane (BPV_MAXK)
            set_error_msg("mixed batch: n_bits*m above 2^16 is not supported by the batch verifiers");
            return BP_EINVAL;
        }
        if (proof_off[i + 1] < proof_off[i] || proof_off[i + 1] - proof_off[i] != (size_t)(wip ? 195 : 292) + 66 * (size_t)k) {
            set_error_msg("mixed batch: proof length inconsistent with its m");
            return BP_EINVAL;
        }
        com_off[i + 1] = com_off[i] + (size_t)m * 64;
        size_t g = 0;
        while (g < groups.size() && groups[g].m != (int)m) g++;
        if (g == groups.size()) groups.push_back(Group{(int)m, k, {}});
        groups[g].idx.push_back(i);
    }
    std::vector<uint8_t> pb, cb, vb;
    for (const Group& g : groups) {
        const size_t cnt = g.idx.size(), plen = (size_t)(wip ? 195 : 292) + 66 * (size_t)g.k, clen = (size_t)g.m * 64;
        pb.resize(cnt * plen);
        cb.resize(cnt * clen);
        vb.resize(cnt);
        for (size_t j = 0; j < cnt; j++) {
            memcpy(&pb[j * plen], proofs + proof_off[g.idx[j]], plen);
            memcpy(&cb[j * clen], coms_xy + com_off[g.idx[j]], clen);
        }
        void *d_p, *d_c, *d_v;
        int rc = stage_batch(ctx, pb.data(), plen, cb.data(), g.m, cnt, &d_p, &d_c, &d_v);
        if (rc != BP_OK) return rc;
        {
            CtxLock lk(ctx);
            rc = verify_batch_device_nolock(ctx, gens, n_bits, g.m, g.k, d_p, plen, d_c, cnt, d_v, wip);
            if (rc != BP_OK) return rc;
            HIPCHK(hipMemcpyAsync(vb.data(), d_v, cnt, hipMemcpyDeviceToHost, ctx->stream), "memcpy verdicts");
            HIPCHK(hipStreamSynchronize(ctx->stream), "sync");
        }
        for (size_t j = 0; j < cnt; j++) verdicts[g.idx[j]] = vb[j];
    }
    return BP_OK;
}
int bp_rangeproof_verify_batch_mixed(bp_ctx* ctx, const bp_gens* gens, int n_bits, const uint8_t* proofs, const size_t* proof_off,
                                     const uint8_t* coms_xy, const uint32_t* m_per_proof, size_t nproofs, uint8_t* verdicts) {
    return verify_batch_mixed(ctx, gens, n_bits, proofs, proof_off, coms_xy, m_per_proof, nproofs, verdicts, 0);
}
int bp_wip_rangeproof_verify_batch_mixed(bp_ctx* ctx, const bp_gens* gens, int n_bits, const uint8_t* proofs, const size_t* proof_off,
                                         const uint8_t* coms_xy, const uint32_t* m_per_proof, size_t nproofs, uint8_t* verdicts) {
    return verify_batch_mixed(ctx, gens, n_bits, proofs, proof_off, coms_xy, m_per_proof, nproofs, verdicts, 1);
}

}  // extern "C"
