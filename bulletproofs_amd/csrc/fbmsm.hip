// Latency-oriented MSM over the resident generator table, for ONE or a few jobs at a time: the shape
// of the prover (A, S, T1, T2, every L_j / R_j of the inner-product rounds:
// /root/reference/src/proofs/range_proof.rs:93-122, 168-169; inner_product.rs:70-115;
// weighted_inner_product.rs:103-154, 222-239) and of a single-proof verifier (range_proof.rs:742-774).
//
// bp_msm_batch's kernel gives 16 lanes to a job because it expects thousands of jobs.  The prover has
// two jobs per round and waits for them, so here a job spreads over the whole chip:
//   job j = sum over its SPARSE generator terms (gidx[t], scalar[t])  +  sum over its extra points
//   k_fb_acc   grid (nblk + nxblk, njobs): blocks < nblk stride over the job's (term, window) items -
//              one table gather + one mixed addition each, no doublings; blocks >= nblk multiply the
//              job-specific points (GLV ladder, four lanes each); LDS tree -> one XYZZ partial per block
//   k_fb_final one block per job: sum of the partials, ONE inversion, canonical x||y bytes
// The scalars of the generator terms arrive as window digits (k_mb_digits of msm_batch.hip).
#include <hip/hip_runtime.h>

#include "ec.h"
#include "ecmul.h"
#include "ecwave.h"
#include "fq.h"
#include "kernels.h"

namespace bp {

#define FB_T 256

__device__ __forceinline__ xyzz fb_block_sum(xyzz acc, xyzz* red, int tid) {
    xyzz_group_tree_quad<FB_T>(acc, tid, 0, red);  // one level of single-lane additions, seven of quad-shared ones (ecwave.h)
    return acc;                                     // valid in thread 0
}

__global__ __launch_bounds__(FB_T) void k_fb_acc(const affine_mem* __restrict__ table, int W, int NB, const tdig_t* __restrict__ dig,
                                                 const uint32_t* __restrict__ gidx, const uint32_t* __restrict__ toff,
                                                 const affine_mem* __restrict__ xpts, const uint8_t* __restrict__ xsc_be,
                                                 const uint32_t* __restrict__ xoff, int nblk, int nxblk, xyzz* __restrict__ vartab,
                                                 xyzz* __restrict__ partial, int pre) {
    __shared__ xyzz red[FB_T];
    const int tid = threadIdx.x, blk = blockIdx.x, job = blockIdx.y;
    xyzz acc = xyzz_inf();
    if (blk < nblk) {
        const uint32_t t0 = toff[job], t1 = toff[job + 1];
        const size_t items = (size_t)(t1 - t0) * W;
        for (size_t it = (size_t)blk * FB_T + tid; it < items; it += (size_t)nblk * FB_T) {
            const uint32_t t = t0 + (uint32_t)(it / W);
            const int w = (int)(it % W);
            int d = dig[(size_t)t * W + w];
            if (d != 0) {
                affine e = affine_load(table[((size_t)gidx[t] * W + w) * NB + ((d < 0 ? -d : d) - 1)]);
                xyzz_madd_s(acc, e, d < 0);
            }
        }
    } else {
        const uint32_t x0 = xoff[job], x1 = xoff[job + 1];
        const int xb = blk - nblk;
        xyzz* tab = vartab + (((size_t)job * nxblk + xb) * FB_T + tid) * ECMUL_NT;
        // four lanes per job-specific point (ec_mul_glv_slice): these calls wait for the result, so the length of the
        // longest lane is what counts.  pre = window ranges per GLV half of tables that are already there (k_fb_pretab: 4, i.e.
        // 2 pre lanes per point - with three ranges a ladder is 44 doublings and 11 additions).
        const int spp = pre ? 2 * pre : 4;
        const uint32_t lane = (uint32_t)(xb * FB_T + tid);  // (pre: one pass, lane = spp * point + slice as in k_fb_pretab)
        const int slice = (int)(lane % spp);
        for (uint32_t v = x0 + lane / spp; v < x1; v += pre ? 0xffffffffu - v : (uint32_t)nxblk * (FB_T / spp)) {
            // pre: the slice's window table is already there (k_fb_pretab, built while the caller derived the scalars)
            const affine Pv = affine_load(xpts[v]);
            const fq kv = fq_from_be_reduce(xsc_be + 32 * (size_t)v);
            xyzz r = !pre ? ec_mul_glv_slice(Pv, kv, slice, tab) : affine_is_inf(Pv) ? xyzz_inf() : ec_slice_ladder(kv, slice, tab, pre);
            xyzz_add(acc, r);
        }
    }
    xyzz s = fb_block_sum(acc, red, tid);
    if (tid == 0) partial[(size_t)job * (nblk + nxblk) + blk] = s;
}

// out_raw != nullptr: the sum itself (XYZZ, 144 bytes) for callers that finish on the host - with a handful of jobs the one
// division-step inversion of a lone lane is ~45 us of this kernel's ~55, and the host (same code, ec.h) does it in ~2 us
__global__ __launch_bounds__(FB_T) void k_fb_final(const xyzz* __restrict__ partial, int nparts, uint8_t* __restrict__ out_be,
                                                   xyzz* __restrict__ out_raw) {
    __shared__ xyzz red[FB_T];
    const int tid = threadIdx.x, job = blockIdx.x;
    xyzz acc = xyzz_inf();
    for (int i = tid; i < nparts; i += FB_T) {
        xyzz o = partial[(size_t)job * nparts + i];
        xyzz_add(acc, o);
    }
    xyzz s = fb_block_sum(acc, red, tid);
    if (tid == 0) {
        if (out_raw)
            out_raw[job] = s;
        else
            affine_to_be64(out_be + 64 * (size_t)job, xyzz_to_affine_var(s));
    }
}

// The scalar-independent half of a single job's job-specific points, ahead of the call: lane 6 p + slice builds the window table
// that k_fb_acc's lane (block (6 p + slice) / FB_T, thread (6 p + slice) % FB_T) of job 0 reads for point p.  FB_PRE_PARTS = 3
// window ranges per GLV half, six lanes per point: built ahead of time the tables cost the call nothing as long as they are ready
// when the scalars are - measured on C1: two ranges 199 us of tables (hidden) + 298 us of ladders, four ranges 361 us (73 us of them
// exposed) + 179 us, three ranges 88 doublings in the tables and 44 doublings + 11 additions in a ladder.  The single-proof verifiers know their proof's points before they know the scalars (two transcript
// hashes, a fixed-base round trip and the whole scalar vector lie in between): the tables are built on the side stream meanwhile.
// Several jobs (the verify / fast_verify calls: eq. 65's right side, P, the inner-product check): blockIdx.y = job, whose points
// are xbe[off[job] .. off[job + 1]) and whose lanes start at block job * gridDim.x, as in k_fb_acc.
__global__ __launch_bounds__(FB_T) void k_fb_pretab(const uint8_t* __restrict__ xbe, FbPreJobs jobs, xyzz* __restrict__ vartab) {
    const int job = blockIdx.y;
    const int L = blockIdx.x * FB_T + threadIdx.x, pl = L / (2 * FB_PRE_PARTS), slice = L % (2 * FB_PRE_PARTS);
    if (pl >= (int)(jobs.off[job + 1] - jobs.off[job])) return;
    const size_t p = jobs.off[job] + (size_t)pl;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(xbe + 64 * p);
    affine_mem m;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        m.x.w[k] = __builtin_bswap32(src[7 - k]);
        m.y.w[k] = __builtin_bswap32(src[15 - k]);
    }
    if (!affine_mem_valid(m, false)) return;  // infinity / not on the curve: the call itself skips or rejects the point
    ec_slice_table(affine_load(m), slice, vartab + (((size_t)job * gridDim.x + blockIdx.x) * FB_T + threadIdx.x) * ECMUL_NT, FB_PRE_PARTS);
}
hipError_t launch_fb_pretab(const uint8_t* xbe, const FbPreJobs& jobs, int nj, int nxblk, xyzz* vartab, hipStream_t st) {
    if (nj <= 0 || nxblk <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_fb_pretab, dim3((unsigned)nxblk, (unsigned)nj), dim3(FB_T), 0, st, xbe, jobs, vartab);
    return hipGetLastError();
}

size_t fb_vartab_bytes(size_t njobs, int nxblk) { return njobs * (size_t)nxblk * FB_T * ECMUL_NT * sizeof(xyzz); }

// nblk: blocks per job over the generator items (>= 1); nxblk: blocks per job over the extras (0 = none)
hipError_t launch_fb_msm(size_t njobs, const affine_mem* table, int W, int NB, const tdig_t* dig, const uint32_t* gidx, const uint32_t* toff,
                         const affine_mem* xpts, const uint8_t* xsc_be, const uint32_t* xoff, int nblk, int nxblk, xyzz* vartab, xyzz* partial,
                         uint8_t* out_be, hipStream_t st, int pre, xyzz* out_raw) {
    if (njobs == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fb_acc, dim3((unsigned)(nblk + nxblk), (unsigned)njobs), dim3(FB_T), 0, st, table, W, NB, dig, gidx, toff, xpts, xsc_be,
                       xoff, nblk, nxblk, vartab, partial, pre);
    hipLaunchKernelGGL(k_fb_final, dim3((unsigned)njobs), dim3(FB_T), 0, st, partial, nblk + nxblk, out_be, out_raw);
    return hipGetLastError();
}

}  // namespace bp
