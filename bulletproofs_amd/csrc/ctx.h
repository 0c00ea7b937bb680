// Internal context of libbpmsm (opaque `bp_ctx` of include/bpmsm.h): one HIP stream plus
// grow-only device workspaces, so steady-state calls allocate nothing.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <mutex>

#include "ec.h"

namespace bp {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);  // BP_OK / BP_ENOMEM / BP_EHIP; contents are NOT preserved on growth
    void release();
};

struct MsmPlan {
    int c, W, T, M, S;
    size_t npad;
};
MsmPlan plan_msm(size_t n, int forced_c);

void set_error(const char* where, hipError_t e);
void set_error_msg(const char* msg);

}  // namespace bp

struct bp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int* d_err = nullptr;
    int forced_c = 0;
    int verify_lanes = 0;
    // optional per-kernel timing (bp_profile_*): HIP events on this context's stream
    bool profiling = false;
    static const int kMaxProf = 8;
    hipEvent_t prof_ev[kMaxProf + 1] = {};
    int prof_n = 0;                 // kernels recorded in the pending call
    int prof_ids[kMaxProf] = {};
    double prof_ms[kMaxProf] = {};  // accumulated per kernel id
    long prof_calls[kMaxProf] = {};
    std::mutex mu;
    bp::DevBuf in_points_be, in_scalars_be, points, digits, partial, wsum, result, initbuf;
    bp::DevBuf scratch[8];
    bp::DevBuf vartab, fixres;
    void release_all() {
        in_points_be.release();
        in_scalars_be.release();
        points.release();
        digits.release();
        partial.release();
        wsum.release();
        result.release();
        initbuf.release();
        for (auto& s : scratch) s.release();
        vartab.release();
        fixres.release();
    }
};

namespace bp {
int msm_on_device(bp_ctx* ctx, const affine_mem* d_points, const uint8_t* d_scalars_be, size_t n, const affine_mem* d_init);
}
