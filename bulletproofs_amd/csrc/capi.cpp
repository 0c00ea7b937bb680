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

// Window width / slicing plan for an n-point MSM (see msm_kernels.hip).
MsmPlan plan_msm(size_t n, int forced_c) {
    MsmPlan pl;
    int c;
    if (const char* e = getenv("BP_MSM_C")) forced_c = atoi(e);  // tuning experiments
    if (forced_c >= 2 && forced_c <= 12)
        c = forced_c;
    else if (n <= 32)
        c = 3;
    else if (n <= 96)
        c = 4;
    else if (n <= 384)
        c = 5;
    else if (n <= 1024)
        c = 6;
    else if (n <= 3072)
        c = 7;
    else if (n <= 12288)
        c = 8;
    else if (n <= 65536)
        c = 9;
    else
        c = 10;  // measured at 2^20 points: c = 10 / 8192-point slices beats c = 11 / 16384 by 8x
    pl.c = c;
    pl.W = 256 / c + 1;
    int NB = 1 << (c - 1);
    pl.T = NB >= 256 ? 256 : (NB >= 128 ? 128 : 64);
    // slice length: enough points per bucket to amortise the in-LDS reduction, enough slices to fill 256 CUs
    size_t M = (size_t)NB * 16;
    if (M < 256) M = 256;
    if (M > 8192) M = 8192;
    while (M > (size_t)NB * 4 && M > 256 && ((n + M - 1) / M) * pl.W < 512) M /= 2;
    if (const char* e = getenv("BP_MSM_M")) M = (size_t)atol(e);
    if (M > n && n > 0) M = n;
    if (M == 0) M = 1;
    pl.M = (int)M;
    pl.S = n ? (int)((n + M - 1) / M) : 1;
    pl.npad = (size_t)pl.S * M;
    return pl;
}

#define HIPCHK(expr, where)              \
    do {                                 \
        hipError_t e__ = (expr);         \
        if (e__ != hipSuccess) {         \
            set_error(where, e__);       \
            return BP_EHIP;              \
        }                                \
    } while (0)

// out = init + sum scalars[i] * points[i]; d_points (limb form) and d_scalars_be on device.
// Leaves the result in ctx->d_result (limb form) and ctx->d_result_be; does not synchronise.
int msm_on_device(bp_ctx* ctx, const affine_mem* d_points, const uint8_t* d_scalars_be, size_t n, const affine_mem* d_init) {
    hipStream_t st = ctx->stream;
    MsmPlan pl = plan_msm(n, ctx->forced_c);
    int rc;
    if ((rc = ctx->digits.reserve((size_t)pl.W * pl.npad * sizeof(int16_t))) != BP_OK) return rc;
    if ((rc = ctx->partial.reserve((size_t)pl.W * pl.S * sizeof(xyzz))) != BP_OK) return rc;
    if ((rc = ctx->wsum.reserve((size_t)pl.W * sizeof(xyzz))) != BP_OK) return rc;
    if ((rc = ctx->result.reserve(sizeof(affine_mem) + 64)) != BP_OK) return rc;
    affine_mem* d_res = (affine_mem*)ctx->result.p;
    uint8_t* d_res_be = (uint8_t*)ctx->result.p + sizeof(affine_mem);
    if (n == 0) {
        // empty sum: result = init (or infinity)
        HIPCHK(hipMemsetAsync(ctx->wsum.p, 0, sizeof(xyzz), st), "memset");
        HIPCHK(launch_msm_horner((const xyzz*)ctx->wsum.p, 1, 1, d_init, d_res, d_res_be, st), "horner");
        return BP_OK;
    }
    if (msm_slice_lds_bytes(pl.c, pl.M, pl.T) > 160 * 1024) {
        set_error_msg("msm plan exceeds LDS");
        return BP_EINVAL;
    }
    HIPCHK(launch_scalars_to_digits(d_scalars_be, n, pl.npad, pl.c, pl.W, (int16_t*)ctx->digits.p, st), "digits");
    HIPCHK(launch_msm_slice(d_points, (const int16_t*)ctx->digits.p, n, pl.npad, pl.M, pl.c, pl.W, pl.S, pl.T, (xyzz*)ctx->partial.p, st), "slice");
    HIPCHK(launch_msm_reduce_slices((const xyzz*)ctx->partial.p, pl.S, pl.W, (xyzz*)ctx->wsum.p, st), "reduce");
    HIPCHK(launch_msm_horner((const xyzz*)ctx->wsum.p, pl.W, pl.c, d_init, d_res, d_res_be, st), "horner");
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
    bp_ctx* ctx = new bp_ctx();
    ctx->device = device;
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
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->release_all();
    if (ctx->d_err) (void)hipFree(ctx->d_err);
    for (int i = 0; i <= bp_ctx::kMaxProf; i++)
        if (ctx->prof_ev[i]) (void)hipEventDestroy(ctx->prof_ev[i]);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

void bp_msm_set_window(bp_ctx* ctx, int c) {
    if (ctx) ctx->forced_c = c;
}

int bp_msm_device(bp_ctx* ctx, const void* d_points_xy, const void* d_scalars_be, size_t n, const uint8_t* init_xy, uint8_t out_xy[64]) {
    if (!ctx || !out_xy || (n && (!d_points_xy || !d_scalars_be))) return BP_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    HIPCHK(hipMemsetAsync(ctx->d_err, 0, sizeof(int), ctx->stream), "memset err");
    int rc;
    if ((rc = ctx->points.reserve((n ? n : 1) * sizeof(affine_mem))) != BP_OK) return rc;
    HIPCHK(launch_ingest_points((const uint8_t*)d_points_xy, (affine_mem*)ctx->points.p, n, ctx->d_err, ctx->stream), "ingest");
    const affine_mem* d_init;
    if ((rc = upload_init(ctx, init_xy, &d_init)) != BP_OK) return rc;
    if ((rc = msm_on_device(ctx, (const affine_mem*)ctx->points.p, (const uint8_t*)d_scalars_be, n, d_init)) != BP_OK) return rc;
    return msm_fetch_result(ctx, out_xy);
}

int bp_msm(bp_ctx* ctx, const uint8_t* points_xy, const uint8_t* scalars_be, size_t n, const uint8_t* init_xy, uint8_t out_xy[64]) {
    if (!ctx || !out_xy || (n && (!points_xy || !scalars_be))) return BP_EINVAL;
    void *d_pts = nullptr, *d_sc = nullptr;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
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
