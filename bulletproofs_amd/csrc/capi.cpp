// C ABI of libbpmsm.so (declared in include/bpmsm.h): context management and the generic MSM.
// Host-side plumbing only: every group operation happens in the HIP kernels; there is no CPU
// fallback — a HIP failure is returned as BP_EHIP.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>

#include "../../include/bpmsm.h"
#include "ctx.h"
#include "kernels.h"

namespace bp {

static thread_local std::string g_last_error;

void set_error(const char* where, hipError_t e) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s: %s", where, hipGetErrorString(e));
    g_last_error = buf;
}
void set_error_msg(const char* msg) { g_last_error = msg; }

int DevBuf::reserve(size_t bytes) {
    if (bytes <= cap) return BP_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        set_error("hipMalloc", e);
        return e == hipErrorOutOfMemory ? BP_ENOMEM : BP_EHIP;
    }
    cap = want;
    return BP_OK;
}
void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

// Number of signed c-bit windows covering a GLV half-scalar (|k| < 2^129) and its recoding carry.
static int glv_windows(int c) { return (129 + c) / c; }

// Window width / slicing plan for the LDS path over n2 = 2n virtual (GLV) points (see msm_kernels.hip).
MsmPlan plan_msm(size_t n2, int forced_c) {
    MsmPlan pl;
    int c;
    if (const char* e = bp_tune("BP_MSM_C")) forced_c = atoi(e);  // tuning experiments
    if (forced_c >= 2 && forced_c <= 12)
        c = forced_c;
    else if (n2 <= 32)
        c = 3;
    else if (n2 <= 96)
        c = 4;
    else if (n2 <= 384)
        c = 5;
    else if (n2 <= 1024)
        c = 6;
    else if (n2 <= 3072)
        c = 7;
    else if (n2 <= 12288)
        c = 8;
    else if (n2 <= 65536)
        c = 9;
    else
        c = 10;
    pl.c = c;
    pl.W = glv_windows(c);
    int NB = 1 << (c - 1);
    pl.T = NB >= 256 ? 256 : (NB >= 128 ? 128 : 64);
    // slice length: enough points per bucket to amortise the in-LDS reduction, enough slices to fill 256 CUs
    size_t M = (size_t)NB * 16;
    if (M < 256) M = 256;
    if (M > 8192) M = 8192;
    while (M > (size_t)NB * 4 && M > 256 && ((n2 + M - 1) / M) * pl.W < 512) M /= 2;
    if (const char* e = bp_tune("BP_MSM_M")) M = (size_t)atol(e);
    if (M > n2 && n2 > 0) M = n2;
    if (M == 0) M = 1;
    pl.M = (int)M;
    pl.S = n2 ? (int)((n2 + M - 1) / M) : 1;
    pl.npad = (size_t)pl.S * M;
    return pl;
}

// Plan of the sorted path (msm_sorted.hip).  Cost model in mixed-add equivalents of whole-GPU
// throughput: W * (n2 + 3 NB) for bucket accumulation + row/column reduction additions, plus
// W * 146000 for the serial Horner step (one window = a full addition and two coordinate conversions
// in a lone wave, ~7.7 us, the time in which the GPU retires ~146k mixed additions): small MSMs are
// latency-bound and want few windows.  c <= 15 (int16 digits, 64 KB LDS histograms).
SortedPlan plan_msm_sorted(size_t n2, int forced_c) {
    SortedPlan pl;
    if (const char* e = bp_tune("BP_MSM_C")) forced_c = atoi(e);
    int c = 0;
    if (forced_c >= 2 && forced_c <= 15) {
        c = forced_c;
    } else {
        double best = 0;
        for (int cc = 6; cc <= 15; cc++) {
            double cost = (double)glv_windows(cc) * ((double)n2 + 3.0 * (double)(1 << (cc - 1)) + 146000.0);
            if (c == 0 || cost < best) {
                best = cost;
                c = cc;
            }
        }
    }
    pl.c = c;
    pl.W = glv_windows(c);
    pl.NB = 1 << (c - 1);
    // chunk of points per histogram workgroup: ~512 workgroups in flight, 4096..65536 points each
    size_t ch = 4096;
    while (ch < 65536 && ((n2 + ch - 1) / ch) * (size_t)pl.W > 512) ch *= 2;
    if (const char* e = bp_tune("BP_MSM_CH")) ch = (size_t)atol(e) >= 1024 ? ((size_t)atol(e) + 7) & ~(size_t)7 : ch;  // read 8 digits at a time
    pl.CH = (int)ch;
    pl.nchunks = (int)((n2 + ch - 1) / ch);
    if (pl.nchunks < 1) pl.nchunks = 1;
    pl.npad = (size_t)pl.nchunks * ch;
    // segment length: ~2 segments per average bucket (the bucket reduction adds every segment's partial sum twice), at least 32 (lanes of a wave then do equal work
    // and k_srt_rowcol2 adds only a few partial sums per bucket)
    // Small MSMs are latency-bound: shorter segments (more lanes, shorter serial chains) while the grid
    // still has fewer than ~64k lanes.
    pl.L = 32;
    while (pl.L < 256 && (size_t)pl.L * 4 <= n2 / (size_t)pl.NB) pl.L *= 2;
    while (pl.L > 4 && (size_t)pl.W * n2 / pl.L < 65536) pl.L /= 2;
    if (const char* e = bp_tune("BP_MSM_L")) pl.L = atoi(e) > 0 ? atoi(e) : 32;
    pl.L = (pl.L + 7) & ~7;  // k_srt_acc reads the sorted entries in aligned 64-byte lines (8 entries)
    pl.nseg = (int)((pl.npad + pl.L - 1) / pl.L);
    pl.Lmin = pl.L < 16 ? pl.L : 16;
    pl.pstride = (int)((pl.npad + pl.Lmin - 1) / pl.Lmin);
    pl.lo_bits = c / 2;  // LO = 2^lo_bits, HI = NB / LO; LO + HI <= 256 for c <= 15
    if (pl.lo_bits > c - 1) pl.lo_bits = c - 1;
    return pl;
}

// Window groups of the pipelined sorted path (top-down).  Measured at 2^20 / 2^22 points (gpurun_out sweeps,
// DESIGN.md section 5): an accumulation launch of >= 3 windows fills the SIMDs; the LAST group is one window, so that the
// exposed tail is one window's bucket reduction and a single addition (its Horner step is done by k_srt_dbl while the
// window still accumulates) - 5,3,1 at W = 9: 2.38-2.51 ms against 2.53-2.65 for 5,4 and 2.9 for 4,2,2,1.
// BP_MSM_GROUPS="3,3,2,1" overrides (tuning).
static int plan_window_groups(int W, int* gsz, size_t n2) {
    int ng = 0;
    if (const char* e = bp_tune("BP_MSM_GROUPS")) {
        int left = W;
        const char* p = e;
        while (*p && left > 0 && ng < SortedStreams::kMaxGroups) {
            int v = atoi(p);
            if (v <= 0) break;
            if (v > left || ng == SortedStreams::kMaxGroups - 1) v = left;
            gsz[ng++] = v;
            left -= v;
            while (*p && *p != ',') p++;
            if (*p == ',') p++;
        }
        if (left > 0 && ng > 0) gsz[ng - 1] += left;
        if (ng > 0) return ng;
    }
    if (W < 4) return 0;
    if (W < 6) {
        gsz[0] = W - W / 2;
        gsz[1] = W / 2;
        return 2;
    }
    // from 2^21 points an accumulation launch is long enough that a middle group only adds a reduction beside the last window:
    // W - 1 windows, then one (2^22 points, W = 9: 8,1 6.87 ms, 4,4,1 6.97, 6,2,1 7.30, 5,3,1 7.55; at 2^20 all of these are within
    // 2 % of 2.17 ms - gpurun_out/r3_groups.log)
    if (n2 >= ((size_t)1 << 22)) {
        gsz[0] = W - 1;
        gsz[1] = 1;
        return 2;
    }
    gsz[0] = ((W - 1) * 5 + 7) / 8;
    gsz[1] = W - 1 - gsz[0];
    gsz[2] = 1;
    return 3;
}

#define HIPCHK(expr, where)              \
    do {                                 \
        hipError_t e__ = (expr);         \
        if (e__ != hipSuccess) {         \
            set_error(where, e__);       \
            return BP_EHIP;              \
        }                                \
    } while (0)

// Window phase of an MSM: wsum[w] = sum of the w-th signed GLV digits times the points, for n >= 1
// terms; leaves W XYZZ window sums in ctx->wsum and reports the window width / count used.
// want_c = 0: planner's choice (and its choice of path); otherwise the sorted path with that width.
// Path: GLV split (2n virtual points, 129-bit scalars) -> LDS slice path (tiny n) or global sorted path.
static int msm_windows_on_device(bp_ctx* ctx, const affine_mem* d_points, const uint8_t* d_scalars_be, size_t n, int want_c, int* c_out,
                                 int* W_out, bool full_msm = false) {
    hipStream_t st = ctx->stream;
    int rc;
    if ((rc = ctx->wsum.reserve(72 * sizeof(xyzz))) != BP_OK) return rc;
    if (n == 0 || n >= ((size_t)1 << 30)) {
        set_error_msg("msm: term count out of range");
        return BP_EINVAL;
    }
    const size_t n2 = 2 * n;
    int path = ctx->msm_path;  // 0 auto, 1 LDS slices, 2 sorted
    if (const char* e = bp_tune("BP_MSM_PATH")) path = atoi(e);
    const bool sort2 = path == 3;  // sorted path with the two-pass sort at any size
    if (path == 3) path = 2;
    if (path == 0) path = n2 >= 256 ? 2 : 1;  // measured: the sorted path wins from 2^7 points up
    if (want_c) path = 2;
    const int forced_c = want_c ? want_c : ctx->forced_c;
    if ((rc = ctx->lam.reserve(n * sizeof(affine_mem))) != BP_OK) return rc;
    const affine_mem* lam = (const affine_mem*)ctx->lam.p;
    if (path == 1) {
        HIPCHK(launch_glv_expand(d_points, (affine_mem*)ctx->lam.p, n, st), "glv expand");
        MsmPlan pl = plan_msm(n2, forced_c);
        *c_out = pl.c;
        *W_out = pl.W;
        if (msm_slice_lds_bytes(pl.c, pl.M, pl.T) > 64 * 1024) {
            set_error_msg("msm plan exceeds LDS");
            return BP_EINVAL;
        }
        if ((rc = ctx->digits.reserve((size_t)pl.W * pl.npad * sizeof(int16_t))) != BP_OK) return rc;
        if ((rc = ctx->partial.reserve((size_t)pl.W * pl.S * sizeof(xyzz))) != BP_OK) return rc;
        HIPCHK(launch_glv_digits(d_scalars_be, n, pl.npad, pl.c, pl.W, (int16_t*)ctx->digits.p, st), "digits");
        HIPCHK(launch_msm_slice(d_points, lam, n, (const int16_t*)ctx->digits.p, n2, pl.npad, pl.M, pl.c, pl.W, pl.S, pl.T, (xyzz*)ctx->partial.p, st),
               "slice");
        HIPCHK(launch_msm_reduce_slices((const xyzz*)ctx->partial.p, pl.S, pl.W, (xyzz*)ctx->wsum.p, st), "reduce");
    } else {
        SortedPlan pl = plan_msm_sorted(n2, forced_c);
        pl.sort2 = sort2 ? 1 : 0;
        *c_out = pl.c;
        *W_out = pl.W;
        const int W = pl.W;
        const size_t NB = pl.NB;
        const size_t nout = ((size_t)1 << pl.lo_bits) + (NB >> pl.lo_bits);
        if ((rc = ctx->digits.reserve((size_t)W * pl.npad * sizeof(int16_t))) != BP_OK) return rc;
        size_t cnt_bytes = (size_t)W * pl.nchunks * NB * 4;
        if (srt2_cnt_words(pl, n2) * 4 > cnt_bytes) cnt_bytes = srt2_cnt_words(pl, n2) * 4;
        if ((rc = ctx->srt[0].reserve(cnt_bytes)) != BP_OK) return rc;
        if ((rc = ctx->srt[1].reserve((size_t)W * (NB + 1) * 4)) != BP_OK) return rc;
        if ((rc = ctx->srt[2].reserve((size_t)W * pl.npad * 8)) != BP_OK) return rc;
        if ((rc = ctx->srt[3].reserve((size_t)W * pl.npad * 8)) != BP_OK) return rc;
        if ((rc = ctx->srt[4].reserve((size_t)W * NB * sizeof(xyzz))) != BP_OK) return rc;
        if ((rc = ctx->srt[5].reserve((size_t)W * pl.pstride * sizeof(xyzz))) != BP_OK) return rc;
        if ((rc = ctx->srt[6].reserve((size_t)W * nout * sizeof(xyzz))) != BP_OK) return rc;
        SortedBufs b;
        b.cnt = (uint32_t*)ctx->srt[0].p;
        b.bstart = (uint32_t*)ctx->srt[1].p;
        b.ent = (uint2*)ctx->srt[2].p;
        b.tmp = (uint2*)ctx->srt[3].p;
        b.Bk = (xyzz*)ctx->srt[4].p;
        b.Pseg = (xyzz*)ctx->srt[5].p;
        b.RC = (xyzz*)ctx->srt[6].p;
        // large MSMs: window groups, top-down; every group's latency-bound tail (bucket reduction, Horner chain)
        // overlaps the following groups' accumulation (not when the caller wants every raw window sum: the
        // multi-GPU partials)
        int gsz[SortedStreams::kMaxGroups];
        int ng = 0;
        // from 2^19 points: below that one group on one stream is as fast or faster (2^16 / 2^17 / 2^18 points: 0.69 / 0.93 / 1.12 ms
        // against 0.75 / 0.97 / 1.16 in groups; 2^19: 1.59 against 1.40; gpurun_out/r3_mid_msm.log).  bp_msm_set_pipeline(ctx, 2)
        // takes the groups from 2^16 points on (tests).
        const size_t group_min = ctx->msm_pipeline == 2 ? ((size_t)1 << 17) : ((size_t)1 << 20);
        if (!want_c && n2 >= group_min && ctx->msm_pipeline != 1) ng = plan_window_groups(W, gsz, n2);
        if (ng > 1 && !ctx->srt_streams) {
            SortedStreams* ss = new SortedStreams();
            memset(ss, 0, sizeof(*ss));
            ctx->srt_streams = ss;
            HIPCHK(ctx->side_stream(&ss->st2), "hipStreamCreate");  // the context's side stream (ctx.h)
            // An ordinary stream, NOT a high-priority one: a priority stream is a hardware queue of its own beside the process's four,
            // and with one more stream in the process (a second context, a prover lane) the cross-queue waits of launch_msm_sorted
            // never completed on this system - the call hung (scripts/experiments/prove_then_msm.py reproduces it with the
            // priority put back).  Its measured gain was nil (2^20 points 2.15 ms either way).
            HIPCHK(hipStreamCreateWithFlags(&ss->st3, hipStreamNonBlocking), "hipStreamCreate");
            ss->st4 = ss->st2;  // a stream of its own shared st3's hardware queue (4 per process) and serialised the two
            for (int g = 0; g < SortedStreams::kMaxGroups; g++) {
                HIPCHK(hipEventCreateWithFlags(&ss->ev_acc[g], hipEventDisableTiming), "hipEventCreate");
                HIPCHK(hipEventCreateWithFlags(&ss->ev_red[g], hipEventDisableTiming), "hipEventCreate");
                HIPCHK(hipEventCreateWithFlags(&ss->ev_sorted[g], hipEventDisableTiming), "hipEventCreate");
            }
            HIPCHK(hipEventCreateWithFlags(&ss->ev_join, hipEventDisableTiming), "hipEventCreate");
            HIPCHK(hipEventCreateWithFlags(&ss->ev_fork, hipEventDisableTiming), "hipEventCreate");
            HIPCHK(hipEventCreateWithFlags(&ss->ev_start, hipEventDisableTiming), "hipEventCreate");
            HIPCHK(hipEventCreateWithFlags(&ss->ev_lam, hipEventDisableTiming), "hipEventCreate");
        }
        if (ng > 1) {
            // lambda * P (one field multiplication per point, 128 MB of traffic at 2^20) on a side stream beside the digit kernel
            SortedStreams* ss = ctx->srt_streams;
            HIPCHK(hipEventRecord(ss->ev_start, st), "hipEventRecord");
            HIPCHK(hipStreamWaitEvent(ss->st2, ss->ev_start, 0), "hipStreamWaitEvent");
            HIPCHK(launch_glv_expand(d_points, (affine_mem*)ctx->lam.p, n, ss->st2), "glv expand");
            HIPCHK(hipEventRecord(ss->ev_lam, ss->st2), "hipEventRecord");
            HIPCHK(launch_glv_digits(d_scalars_be, n, pl.npad, pl.c, pl.W, (int16_t*)ctx->digits.p, st), "digits");
            HIPCHK(hipStreamWaitEvent(st, ss->ev_lam, 0), "hipStreamWaitEvent");
        } else {
            HIPCHK(launch_glv_expand(d_points, (affine_mem*)ctx->lam.p, n, st), "glv expand");
            HIPCHK(launch_glv_digits(d_scalars_be, n, pl.npad, pl.c, pl.W, (int16_t*)ctx->digits.p, st), "digits");
        }
        pl.ncu = ctx->ncu;
        pl.weighted = full_msm && ng <= 1;  // (partials for another rank's combine stay plain window sums)
        int c_eff = pl.c;
        HIPCHK(launch_msm_sorted(d_points, lam, n, (const int16_t*)ctx->digits.p, n2, pl, b, (xyzz*)ctx->wsum.p, st, ctx->srt_streams, gsz, ng, W_out,
                                 &c_eff),
               "sorted msm");
        *c_out = c_eff;
    }
    return BP_OK;
}

// out = init + sum scalars[i] * points[i]; d_points (limb form) and d_scalars_be on device.
// Leaves the result in ctx->d_result (limb form) and ctx->d_result_be; does not synchronise.
int msm_on_device(bp_ctx* ctx, const affine_mem* d_points, const uint8_t* d_scalars_be, size_t n, const affine_mem* d_init) {
    hipStream_t st = ctx->stream;
    int rc;
    if ((rc = ctx->wsum.reserve(72 * sizeof(xyzz))) != BP_OK) return rc;
    if ((rc = ctx->result.reserve(sizeof(affine_mem) + 64 + sizeof(xyzz) + 16)) != BP_OK) return rc;
    affine_mem* d_res = (affine_mem*)ctx->result.p;
    uint8_t* d_res_be = (uint8_t*)ctx->result.p + sizeof(affine_mem);
    const bool want_raw = ctx->msm_want_raw;
    ctx->msm_want_raw = false;
    ctx->msm_result_raw = false;
    if (n == 0) {
        // empty sum: result = init (or infinity)
        HIPCHK(hipMemsetAsync(ctx->wsum.p, 0, sizeof(xyzz), st), "memset");
        HIPCHK(launch_msm_horner((const xyzz*)ctx->wsum.p, 1, 1, d_init, d_res, d_res_be, st), "horner");
        return BP_OK;
    }
    // tiny MSMs: four lanes per point in one workgroup (k_msm_tiny); forced paths / window widths keep the general pipeline
    if (n <= msm_tiny_max() && ctx->msm_path == 0 && ctx->forced_c == 0 && !bp_tune("BP_MSM_PATH") && !bp_tune("BP_MSM_C")) {
        if ((rc = ctx->partial.reserve(msm_tiny_scratch_bytes())) != BP_OK) return rc;
        HIPCHK(launch_msm_tiny(d_points, d_scalars_be, n, d_init, (xyzz*)ctx->partial.p, d_res, d_res_be, st), "msm tiny");
        return BP_OK;
    }
    int c = 0, W = 0;
    if ((rc = msm_windows_on_device(ctx, d_points, d_scalars_be, n, 0, &c, &W, true)) != BP_OK) return rc;
    xyzz* d_raw = (xyzz*)(d_res_be + 64);
    HIPCHK(launch_msm_horner((const xyzz*)ctx->wsum.p, W, c, d_init, d_res, d_res_be, st, want_raw ? d_raw : nullptr, ctx->d_err), "horner");
    ctx->msm_result_raw = want_raw;
    return BP_OK;
}

// Upload an optional init point; returns device pointer (or nullptr) in *d_init.
static int upload_init(bp_ctx* ctx, const uint8_t* init_xy, const affine_mem** d_init) {
    *d_init = nullptr;
    if (!init_xy) return BP_OK;
    int rc;
    if ((rc = ctx->initbuf.reserve(64 + sizeof(affine_mem))) != BP_OK) return rc;
    HIPCHK(hipMemcpyAsync(ctx->initbuf.p, init_xy, 64, hipMemcpyHostToDevice, ctx->stream), "memcpy init");
    affine_mem* d = (affine_mem*)((uint8_t*)ctx->initbuf.p + 64);
    HIPCHK(launch_ingest_points((const uint8_t*)ctx->initbuf.p, d, 1, ctx->d_err, ctx->stream), "ingest init");
    *d_init = d;
    return BP_OK;
}

int msm_fetch_result(bp_ctx* ctx, uint8_t out_xy[64]) {
    int h_err = 0;
    if (ctx->msm_result_raw) {  // XYZZ sum + flag, one copy; the inversion on the host (ec.h: the code k_msm_horner runs otherwise)
        ctx->msm_result_raw = false;
        uint8_t raw[sizeof(xyzz) + sizeof(int)];
        HIPCHK(hipMemcpyAsync(raw, (uint8_t*)ctx->result.p + sizeof(affine_mem) + 64, sizeof(raw), hipMemcpyDeviceToHost, ctx->stream), "memcpy out");
        HIPCHK(hipStreamSynchronize(ctx->stream), "sync");
        memcpy(&h_err, raw + sizeof(xyzz), sizeof(int));
        if (h_err) {
            set_error_msg("input point not on curve");
            return BP_EPOINT;
        }
        xyzz sum;
        memcpy(&sum, raw, sizeof(xyzz));
        affine_to_be64(out_xy, xyzz_to_affine_var(sum));
        return BP_OK;
    }
    uint8_t tmp[64];
    HIPCHK(hipMemcpyAsync(tmp, (uint8_t*)ctx->result.p + sizeof(affine_mem), 64, hipMemcpyDeviceToHost, ctx->stream), "memcpy out");
    HIPCHK(hipMemcpyAsync(&h_err, ctx->d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream), "memcpy err");
    HIPCHK(hipStreamSynchronize(ctx->stream), "sync");
    if (h_err) {
        set_error_msg("input point not on curve");
        return BP_EPOINT;
    }
    memcpy(out_xy, tmp, 64);
    return BP_OK;
}

}  // namespace bp

using namespace bp;

extern "C" {

const char* bp_last_error(void) { return g_last_error.c_str(); }

int bp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int bp_ctx_create(int device, bp_ctx** out) {
    if (!out) return BP_EINVAL;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        set_error_msg("no HIP device available (libbpmsm has no CPU fallback)");
        return BP_EHIP;
    }
    if (device < 0 || device >= ndev) {
        set_error_msg("device index out of range");
        return BP_EINVAL;
    }
    HIPCHK(hipSetDevice(device), "hipSetDevice");
    if (int rc = noalloc_init_device(device)) return rc;
    bp_ctx* ctx = new bp_ctx();
    ctx->device = device;
    if (hipDeviceGetAttribute(&ctx->ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) ctx->ncu = 0;
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error("hipStreamCreate", e);
        delete ctx;
        return BP_EHIP;
    }
    e = hipMalloc((void**)&ctx->d_err, 256);
    if (e != hipSuccess) {
        set_error("hipMalloc", e);
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return BP_EHIP;
    }
    (void)hipMemsetAsync(ctx->d_err, 0, 256, ctx->stream);
    *out = ctx;
    return BP_OK;
}

void bp_ctx_destroy(bp_ctx* ctx) {
    if (!ctx) return;
    for (bp_ctx* sub : ctx->prove_lanes) bp_ctx_destroy(sub);
    ctx->prove_lanes.clear();
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->vside) (void)hipStreamSynchronize(ctx->vside);  // the verifier's side lane may still be running
    ctx->release_all();
    if (ctx->d_err) (void)hipFree(ctx->d_err);
    for (int i = 0; i <= bp_ctx::kMaxProf; i++)
        if (ctx->prof_ev[i]) (void)hipEventDestroy(ctx->prof_ev[i]);
    if (SortedStreams* ss = ctx->srt_streams) {
        if (ss->st3) (void)hipStreamSynchronize(ss->st3);
        if (ss->st3) (void)hipStreamDestroy(ss->st3);  // (st2 is the context's side stream, destroyed below)
        for (int g = 0; g < SortedStreams::kMaxGroups; g++) {
            if (ss->ev_acc[g]) (void)hipEventDestroy(ss->ev_acc[g]);
            if (ss->ev_red[g]) (void)hipEventDestroy(ss->ev_red[g]);
            if (ss->ev_sorted[g]) (void)hipEventDestroy(ss->ev_sorted[g]);
        }
        if (ss->ev_join) (void)hipEventDestroy(ss->ev_join);
        if (ss->ev_fork) (void)hipEventDestroy(ss->ev_fork);
        if (ss->ev_start) (void)hipEventDestroy(ss->ev_start);
        if (ss->ev_lam) (void)hipEventDestroy(ss->ev_lam);
        delete ss;
        ctx->srt_streams = nullptr;
    }
    if (ctx->vside) {
        (void)hipStreamSynchronize(ctx->vside);
        (void)hipStreamDestroy(ctx->vside);
    }
    if (ctx->v_ev_side) (void)hipEventDestroy(ctx->v_ev_side);
    if (ctx->v_ev_fork) (void)hipEventDestroy(ctx->v_ev_fork);
    if (ctx->v_ev_join) (void)hipEventDestroy(ctx->v_ev_join);
    if (ctx->ev_fbpre) (void)hipEventDestroy(ctx->ev_fbpre);
    if (ctx->ev_fbpre0) (void)hipEventDestroy(ctx->ev_fbpre0);
    if (!ctx->stream_borrowed) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

void bp_msm_set_window(bp_ctx* ctx, int c) {
    if (ctx) ctx->forced_c = c;
}
void bp_msm_set_path(bp_ctx* ctx, int path) {
    if (ctx) ctx->msm_path = (path >= 1 && path <= 3) ? path : 0;
}
void bp_msm_set_pipeline(bp_ctx* ctx, int mode) {
    if (ctx) ctx->msm_pipeline = (mode == 1 || mode == 2) ? mode : 0;
}
void bp_proto_set_path(bp_ctx* ctx, int path) {
    if (ctx) ctx->proto_path = path == 1 ? 1 : 0;
}

int bp_msm_device(bp_ctx* ctx, const void* d_points_xy, const void* d_scalars_be, size_t n, const uint8_t* init_xy, uint8_t out_xy[64]) {
    if (!ctx || !out_xy || (n && (!d_points_xy || !d_scalars_be))) return BP_EINVAL;
    CtxLock lk(ctx);
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    HIPCHK(hipMemsetAsync(ctx->d_err, 0, sizeof(int), ctx->stream), "memset err");
    int rc;
    if ((rc = ctx->points.reserve((n ? n : 1) * sizeof(affine_mem))) != BP_OK) return rc;
    HIPCHK(launch_ingest_points((const uint8_t*)d_points_xy, (affine_mem*)ctx->points.p, n, ctx->d_err, ctx->stream), "ingest");
    const affine_mem* d_init;
    if ((rc = upload_init(ctx, init_xy, &d_init)) != BP_OK) return rc;
    ctx->msm_want_raw = true;
    if ((rc = msm_on_device(ctx, (const affine_mem*)ctx->points.p, (const uint8_t*)d_scalars_be, n, d_init)) != BP_OK) return rc;
    return msm_fetch_result(ctx, out_xy);
}

int bp_msm_auto_window(size_t n) { return n ? plan_msm_sorted(2 * n, 0).c : 0; }

size_t bp_msm_partial_bytes(int c) { return (c >= 2 && c <= 15) ? (size_t)glv_windows(c) * sizeof(xyzz) : 0; }

// window sums of one shard into ctx->wsum (device), asynchronous on ctx->stream; n >= 1
static int msm_partial_launch(bp_ctx* ctx, const void* d_points_xy, const void* d_scalars_be, size_t n, int c) {
    HIPCHK(hipMemsetAsync(ctx->d_err, 0, sizeof(int), ctx->stream), "memset err");
    int rc;
    if ((rc = ctx->points.reserve(n * sizeof(affine_mem))) != BP_OK) return rc;
    HIPCHK(launch_ingest_points((const uint8_t*)d_points_xy, (affine_mem*)ctx->points.p, n, ctx->d_err, ctx->stream), "ingest");
    int cu = 0, W = 0;
    return msm_windows_on_device(ctx, (const affine_mem*)ctx->points.p, (const uint8_t*)d_scalars_be, n, c, &cu, &W);
}

int bp_msm_partial_device(bp_ctx* ctx, const void* d_points_xy, const void* d_scalars_be, size_t n, int c, uint8_t* partial_out) {
    if (!ctx || !partial_out || c < 2 || c > 15 || (n && (!d_points_xy || !d_scalars_be))) return BP_EINVAL;
    CtxLock lk(ctx);
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    const size_t bytes = bp_msm_partial_bytes(c);
    if (n == 0) {  // empty shard: every window sum is the identity (ZZ = 0)
        xyzz inf = xyzz_inf();
        for (int w = 0; w < glv_windows(c); w++) memcpy(partial_out + (size_t)w * sizeof(xyzz), &inf, sizeof(xyzz));
        return BP_OK;
    }
    int rc;
    if ((rc = msm_partial_launch(ctx, d_points_xy, d_scalars_be, n, c)) != BP_OK) return rc;
    int h_err = 0;
    HIPCHK(hipMemcpyAsync(partial_out, ctx->wsum.p, bytes, hipMemcpyDeviceToHost, ctx->stream), "memcpy partial");
    HIPCHK(hipMemcpyAsync(&h_err, ctx->d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream), "memcpy err");
    HIPCHK(hipStreamSynchronize(ctx->stream), "sync");
    if (h_err) {
        set_error_msg("input point not on curve");
        return BP_EPOINT;
    }
    return BP_OK;
}

// The same with the partial LEFT IN DEVICE MEMORY and no synchronisation: the exchange step of a multi-GPU MSM launches
// its all-gather right behind these kernels (bp_ctx_record_event bridges the context's stream to the collective's).
// An off-curve input point is reported by the next bp_ctx_sync / bp_msm_combine_device of this context (BP_EPOINT).
int bp_msm_partial_device_out(bp_ctx* ctx, const void* d_points_xy, const void* d_scalars_be, size_t n, int c, void* d_partial_out) {
    if (!ctx || !d_partial_out || c < 2 || c > 15 || (n && (!d_points_xy || !d_scalars_be))) return BP_EINVAL;
    CtxLock lk(ctx);
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    const size_t bytes = bp_msm_partial_bytes(c);
    if (n == 0) {  // identity window sums: ZZ = 0 (all-zero bytes are an identity, ec.h)
        HIPCHK(hipMemsetAsync(d_partial_out, 0, bytes, ctx->stream), "memset partial");
        return BP_OK;
    }
    int rc;
    if ((rc = msm_partial_launch(ctx, d_points_xy, d_scalars_be, n, c)) != BP_OK) return rc;
    HIPCHK(hipMemcpyAsync(d_partial_out, ctx->wsum.p, bytes, hipMemcpyDeviceToDevice, ctx->stream), "memcpy partial");
    HIPCHK(launch_or_flag(ctx->d_err + 1, ctx->d_err, ctx->stream), "park flag");
    ctx->async_point_check = true;
    return BP_OK;
}

int bp_msm_partial(bp_ctx* ctx, const uint8_t* points_xy, const uint8_t* scalars_be, size_t n, int c, uint8_t* partial_out) {
    if (!ctx || !partial_out || (n && (!points_xy || !scalars_be))) return BP_EINVAL;
    void *d_pts = nullptr, *d_sc = nullptr;
    if (n) {
        CtxLock lk(ctx);
        HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
        int rc;
        if ((rc = ctx->in_points_be.reserve(n * 64)) != BP_OK) return rc;
        if ((rc = ctx->in_scalars_be.reserve(n * 32)) != BP_OK) return rc;
        HIPCHK(hipMemcpyAsync(ctx->in_points_be.p, points_xy, n * 64, hipMemcpyHostToDevice, ctx->stream), "memcpy points");
        HIPCHK(hipMemcpyAsync(ctx->in_scalars_be.p, scalars_be, n * 32, hipMemcpyHostToDevice, ctx->stream), "memcpy scalars");
        d_pts = ctx->in_points_be.p;
        d_sc = ctx->in_scalars_be.p;
    }
    return bp_msm_partial_device(ctx, d_pts, d_sc, n, c, partial_out);
}

static int msm_combine_impl(bp_ctx* ctx, const void* partials, bool on_device, size_t nparts, int c, const uint8_t* init_xy, uint8_t out_xy[64]) {
    if (!ctx || !out_xy || c < 2 || c > 15 || (nparts && !partials) || nparts > 4096) return BP_EINVAL;
    CtxLock lk(ctx);
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    const int W = glv_windows(c);
    int rc;
    if ((rc = ctx->wsum.reserve(72 * sizeof(xyzz))) != BP_OK) return rc;
    if ((rc = ctx->result.reserve(sizeof(affine_mem) + 64)) != BP_OK) return rc;
    if ((rc = ctx->partial.reserve((nparts ? nparts : 1) * (size_t)W * sizeof(xyzz))) != BP_OK) return rc;
    // an off-curve point of an earlier bp_msm_partial_device_out of this context (its flag was parked in d_err[1])
    int h_async = 0;
    if (ctx->async_point_check) HIPCHK(hipMemcpyAsync(&h_async, ctx->d_err + 1, sizeof(int), hipMemcpyDeviceToHost, ctx->stream), "memcpy err");
    HIPCHK(hipMemsetAsync(ctx->d_err, 0, sizeof(int), ctx->stream), "memset err");
    const affine_mem* d_init;
    if ((rc = upload_init(ctx, init_xy, &d_init)) != BP_OK) return rc;
    const xyzz* d_parts = (const xyzz*)partials;
    if (nparts && !on_device) {
        HIPCHK(hipMemcpyAsync(ctx->partial.p, partials, nparts * (size_t)W * sizeof(xyzz), hipMemcpyHostToDevice, ctx->stream), "memcpy partials");
        d_parts = (const xyzz*)ctx->partial.p;
    }
    // partial[g][w] -> wsum[w]: the slice reduction kernel with "slices" strided by W
    HIPCHK(launch_msm_reduce_parts(d_parts, (int)nparts, W, (xyzz*)ctx->wsum.p, ctx->stream), "reduce parts");
    HIPCHK(launch_msm_horner((const xyzz*)ctx->wsum.p, W, c, d_init, (affine_mem*)ctx->result.p, (uint8_t*)ctx->result.p + sizeof(affine_mem),
                             ctx->stream),
           "horner");
    rc = msm_fetch_result(ctx, out_xy);
    if (ctx->async_point_check) {
        ctx->async_point_check = false;
        (void)hipMemsetAsync(ctx->d_err + 1, 0, sizeof(int), ctx->stream);
        if (rc == BP_OK && h_async) {
            set_error_msg("input point not on curve (bp_msm_partial_device_out)");
            return BP_EPOINT;
        }
    }
    return rc;
}
int bp_msm_combine(bp_ctx* ctx, const uint8_t* partials, size_t nparts, int c, const uint8_t* init_xy, uint8_t out_xy[64]) {
    return msm_combine_impl(ctx, partials, false, nparts, c, init_xy, out_xy);
}
int bp_msm_combine_device(bp_ctx* ctx, const void* d_partials, size_t nparts, int c, const uint8_t* init_xy, uint8_t out_xy[64]) {
    return msm_combine_impl(ctx, d_partials, true, nparts, c, init_xy, out_xy);
}

int bp_ctx_record_event(bp_ctx* ctx, void* hip_event) {
    if (!ctx || !hip_event) return BP_EINVAL;
    CtxLock lk(ctx);  // (orders the context's stream after the verifier's side lane first)
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    HIPCHK(hipEventRecord((hipEvent_t)hip_event, ctx->stream), "hipEventRecord");
    return BP_OK;
}

int bp_msm(bp_ctx* ctx, const uint8_t* points_xy, const uint8_t* scalars_be, size_t n, const uint8_t* init_xy, uint8_t out_xy[64]) {
    if (!ctx || !out_xy || (n && (!points_xy || !scalars_be))) return BP_EINVAL;
    void *d_pts = nullptr, *d_sc = nullptr;
    {
        CtxLock lk(ctx);
        HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
        int rc;
        if ((rc = ctx->in_points_be.reserve((n ? n : 1) * 64)) != BP_OK) return rc;
        if ((rc = ctx->in_scalars_be.reserve((n ? n : 1) * 32)) != BP_OK) return rc;
        if (n) {
            HIPCHK(hipMemcpyAsync(ctx->in_points_be.p, points_xy, n * 64, hipMemcpyHostToDevice, ctx->stream), "memcpy points");
            HIPCHK(hipMemcpyAsync(ctx->in_scalars_be.p, scalars_be, n * 32, hipMemcpyHostToDevice, ctx->stream), "memcpy scalars");
        }
        d_pts = ctx->in_points_be.p;
        d_sc = ctx->in_scalars_be.p;
    }
    return bp_msm_device(ctx, d_pts, d_sc, n, init_xy, out_xy);
}

}  // extern "C"
