// Internal context of libbpmsm (opaque `bp_ctx` of include/bpmsm.h): one HIP stream plus
// grow-only device workspaces, so steady-state calls allocate nothing.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <mutex>
#include <vector>

#include "ec.h"
#include "kernels.h"

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
MsmPlan plan_msm(size_t n2, int forced_c);

void set_error(const char* where, hipError_t e);
void set_error_msg(const char* msg);

}  // namespace bp

struct bp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // large MSMs: window groups' bucket reduction (st3) and Horner chain (st2) overlap the next groups' accumulation
    bp::SortedStreams* srt_streams = nullptr;
    int* d_err = nullptr;  // [0] input-validation flag of the current call, [1] parked flag of bp_msm_partial_device_out
    bool async_point_check = false;  // d_err[1] is pending: bp_ctx_sync / bp_msm_combine_device report it
    int forced_c = 0;
    // large MSMs whose result the host reads (bp_msm, bp_msm_device): msm_want_raw asks msm_on_device for the XYZZ sum + flag in
    // one buffer (k_msm_horner's out_raw), msm_result_raw says the last call delivered it that way (msm_fetch_result)
    bool msm_want_raw = false, msm_result_raw = false;
    int ncu = 0;  // compute units of the device (window-group launches are sized per CU)
    int msm_path = 0;  // 0 auto, 1 LDS slice path, 2 sorted path (bp_msm_set_path)
    int msm_pipeline = 0;  // 0 = window groups on side streams from 2^19 points (lowest latency of one call), 1 = one group, 2 = groups from 2^16 points (bp_msm_set_pipeline)
    int verify_lanes = 0;
    // Batch verifier, asynchronous device entry points (bp_verify_set_overlap, verify_api.cpp): successive calls alternate
    // between ctx->stream with the context's workspaces and a side lane (own stream, own workspaces `vws`), so that two
    // batches are in flight and one's latency-bound kernels and kernel tails are filled by the other's work.  Every other
    // entry point orders ctx->stream after the side lane first (CtxLock below).
    int verify_overlap = 1;
    // ONE side stream per context, shared by the verifier's side lane and the sorted MSM's sort / Horner stream (st2): a process
    // has four hardware queues, and a fifth stream lands on an occupied queue and serialises with it (measured: the 2^20-point
    // MSM 2.3 -> 3.0 ms with a separate verifier stream created first).  Work of the two users never depends on each other.
    hipStream_t vside = nullptr;
    hipError_t side_stream(hipStream_t* out) {
        if (!vside) {
            hipError_t e = hipStreamCreateWithFlags(&vside, hipStreamNonBlocking);
            if (e != hipSuccess) return e;
        }
        *out = vside;
        return hipSuccess;
    }
    hipEvent_t v_ev_side = nullptr;
    hipEvent_t v_ev_fork = nullptr, v_ev_join = nullptr;  // one call alone: k_bpv_fix on the side stream beside the proof-specific kernel
    bool v_side_pending = false;
    unsigned v_calls = 0;
    bp::DevBuf vws[8];
    void join_side() {
        if (v_side_pending) {
            (void)hipSetDevice(device);
            (void)hipStreamWaitEvent(stream, v_ev_side, 0);
            v_side_pending = false;
        }
    }
    std::vector<bp_ctx*> prove_lanes;  // batch provers: further contexts, one per lane in flight (proto.cpp), owned by this one
    bool stream_borrowed = false;      // a prover lane whose `stream` is its owner's side stream (not destroyed with the lane)
    int proto_path = 0;  // protocol layer: 0 = every generator MSM from the fixed-base table, 1 = reference-structured (bp_proto_set_path)
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
    bp::DevBuf lam;     // lambda(P_i) = (beta x_i, y_i), the GLV images of the current MSM's points
    bp::DevBuf srt[7];  // sorted-path workspaces (msm_sorted.hip)
    bp::DevBuf scratch[8];
    bp::DevBuf vartab, fixres, fixpart;
    bp::DevBuf proto[6];  // protocol layer (proto.cpp): folded bases, hi_tag, staging
    bp::DevBuf rlc[3];    // random-linear-combination batch verifier (rlc_kernels.hip): weights, MSM scalars, misc
    bp::DevBuf fb[3];     // fixed-table MSM of the protocol layer (fbmsm.hip): inputs, digits, partials + output
    // single-proof verifiers: window tables of the proof's points built on the side stream ahead of the MSM (proto.cpp FbPretab)
    bp::DevBuf fbpre;
    void* h_fbpre = nullptr;                               // pinned staging of the points (64 KiB, allocated at first use)
    hipEvent_t ev_fbpre = nullptr, ev_fbpre0 = nullptr;    // tables built (side stream) / ctx->stream's earlier work done
    int fbpre_n = 0;  // points whose tables are pending in ctx->vartab (0 = none)
    int fbpre_nj = 0;               // ... of this many jobs,
    uint32_t fbpre_off[9] = {};     // job j owning points fbpre_off[j] .. fbpre_off[j + 1)
    void* h_pin = nullptr;  // pinned host staging for the protocol layer's small uploads / read-backs (grow-only)
    size_t h_pin_cap = 0;
    void release_all() {
        in_points_be.release();
        in_scalars_be.release();
        points.release();
        digits.release();
        partial.release();
        wsum.release();
        result.release();
        initbuf.release();
        lam.release();
        for (auto& s : srt) s.release();
        for (auto& s : scratch) s.release();
        for (auto& s : vws) s.release();
        vartab.release();
        fixres.release();
        fixpart.release();
        for (auto& s : proto) s.release();
        for (auto& s : fb) s.release();
        fbpre.release();
        if (h_fbpre) (void)hipHostFree(h_fbpre);
        h_fbpre = nullptr;
        for (auto& s : rlc) s.release();
        if (h_pin) (void)hipHostFree(h_pin);
        h_pin = nullptr;
        h_pin_cap = 0;
    }
};

// Lock of an entry point: the context's mutex, then ctx->stream ordered after the batch verifier's side lane (a no-op unless
// an overlapped verification is pending there).
struct CtxLock {
    std::unique_lock<std::mutex> lk;
    explicit CtxLock(bp_ctx* ctx) : lk(ctx->mu) { ctx->join_side(); }
};

// Opaque `bp_gens` of include/bpmsm.h: generators in limb form + fixed-base window table in HBM.
struct bp_gens {
    int device = 0;
    size_t nm = 0;
    int np = 0;  // 2nm + 2
    // table row of the secp256k1 generator, the base of y*G (`Point::generator() * &y`, range_proof.rs:570-571): row 2nm
    // when the caller's G is the generator (every reference test and bench), else an extra row np after H
    int ibase = 0;
    int nrows = 0;  // table rows: np, or np + 1 with the extra generator row
    int c = 0, W = 0, NB = 0;
    bp::affine_mem* d_pts = nullptr;    // g_vec, h_vec, G, H
    bp::affine_mem* d_table = nullptr;  // [np (+1 when ibase == np)][W][NB]
    size_t table_bytes = 0;
    std::vector<uint8_t> host_be;       // np * 64 bytes (x||y big-endian), for host-side transcript hashing
};

namespace bp {
int msm_fetch_result(bp_ctx* ctx, uint8_t out_xy[64]);
int msm_on_device(bp_ctx* ctx, const affine_mem* d_points, const uint8_t* d_scalars_be, size_t n, const affine_mem* d_init);
int noalloc_init_device(int device);  // verify_api.cpp: resolves this device's copy of the no-allocation scratch array
}
