// Large-N Pippenger MSM for secp256k1 on CDNA4 (gfx950): the raw-MSM configuration of the reference's
// fold idiom (/root/reference/src/proofs/range_proof.rs:770-774 at 2^16..2^22 terms).
//
// The LDS path of msm_kernels.hip gives every (slice, window) workgroup its own bucket set, so the
// bucket-reduction additions grow with the slice count and one workgroup per CU leaves the SIMDs at
// one wave.  Here each window has ONE bucket set in HBM and the work is made uniform per lane:
//
//   k_srt_hist     (chunk, window) workgroups: LDS histogram of |digit|, written per chunk (no global atomics)
//   k_srt_colscan  exclusive scan over the chunks of every bucket count, bucket sizes out
//   k_srt_bstart   per window: exclusive scan of the bucket sizes -> bucket start offsets
//   k_srt_scatter  (chunk, window): LDS cursors; writes the window's points sorted by bucket
//                  (8-byte entry = virtual point index | sign bit, bucket id)
//   k_srt_acc      one lane per SEGMENT of L consecutive sorted entries (not per bucket): every lane
//                  performs exactly L mixed additions whatever the bucket sizes are.  A run that starts
//                  inside the segment is the beginning of its bucket and is written to Bk[bucket]; the
//                  run at the segment start goes to Pseg[segment].
//   k_srt_rowcol2  bucket index b = hi*LO + lo: row sums R_hi and column sums C_lo, one workgroup each, taken
//                  directly from Bk and Pseg (a sum over buckets needs their partials in no particular order,
//                  so there is no per-bucket finalize pass)
//   k_srt_window   sum_b (b+1) B_b = sum_lo lo*C_lo + sum_hi (LO*hi + 1) R_hi by an in-LDS suffix scan
//
// No atomics on group elements; the order of additions inside a bucket depends on the LDS cursor
// race in k_srt_scatter, which cannot change the (canonical, affine) result.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include "ec.h"
#include "ecq.h"
#include "kernels.h"

namespace bp {

// 512-lane workgroups (two waves per SIMD, <= 64 VGPRs): a sort workgroup of a later window group fits beside the two
// accumulation waves per SIMD of the group before it (launch_msm_sorted)
#define SRT_T 512

// Digits are read eight at a time (one 16-byte load per lane and step: a chunk of 65536 digits is 8 dependent
// load -> LDS-atomic steps per lane instead of 64); the columns [n2, npad) hold zero digits, so whole chunks are read.
__device__ __forceinline__ int srt_digit(const uint4& q, int h) {
    const uint32_t wd = (h >> 1) == 0 ? q.x : (h >> 1) == 1 ? q.y : (h >> 1) == 2 ? q.z : q.w;
    return (int)(int16_t)((h & 1) ? (wd >> 16) : (wd & 0xffffu));
}

__global__ __launch_bounds__(SRT_T) void k_srt_hist(const int16_t* __restrict__ digits, size_t n2, size_t npad, int CH, int NB, int w0,
                                                    uint32_t* __restrict__ cnt) {
    extern __shared__ uint32_t srt_h[];
    const int tid = threadIdx.x, chunk = blockIdx.x, w = w0 + blockIdx.y, nchunks = gridDim.x;
    for (int b = tid; b < NB; b += SRT_T) srt_h[b] = 0;
    __syncthreads();
    const size_t base = (size_t)chunk * CH;
    const uint4* dg = reinterpret_cast<const uint4*>(digits + (size_t)w * npad + base);
    if (base < n2) {
        for (int i = tid; i < CH / 8; i += SRT_T) {
            const uint4 q = dg[i];
#pragma unroll
            for (int h = 0; h < 8; h++) {
                const int d = srt_digit(q, h);
                if (d != 0) atomicAdd(&srt_h[(d < 0 ? -d : d) - 1], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t* o = cnt + ((size_t)w * nchunks + chunk) * NB;
    for (int b = tid; b < NB; b += SRT_T) o[b] = srt_h[b];
}

// cnt[w][chunk][b] -> exclusive prefix over chunks (in place); sz[w][b] = bucket size
__global__ void k_srt_colscan(uint32_t* __restrict__ cnt, int nchunks, int NB, int w0, uint32_t* __restrict__ sz) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x, w = w0 + blockIdx.y;
    if (b >= NB) return;
    uint32_t tot = 0;
    for (int ch = 0; ch < nchunks; ch++) {
        uint32_t* p = cnt + ((size_t)w * nchunks + ch) * NB + b;
        uint32_t v = *p;
        *p = tot;
        tot += v;
    }
    sz[(size_t)w * (NB + 1) + b] = tot;
}

// in place: sizes -> exclusive starts, entry NB = number of non-zero digits of the window
__global__ __launch_bounds__(SRT_T) void k_srt_bstart(uint32_t* __restrict__ bstart, int NB, int w0) {
    __shared__ uint32_t ts[SRT_T];
    const int tid = threadIdx.x, w = w0 + blockIdx.x;
    uint32_t* a = bstart + (size_t)w * (NB + 1);
    const int per = (NB + SRT_T - 1) / SRT_T;
    int lo = tid * per, hi = lo + per;
    if (lo > NB) lo = NB;
    if (hi > NB) hi = NB;
    uint32_t s = 0;
    for (int b = lo; b < hi; b++) s += a[b];
    ts[tid] = s;
    __syncthreads();
    for (int off = 1; off < SRT_T; off <<= 1) {
        uint32_t v = ts[tid];
        uint32_t u = tid >= off ? ts[tid - off] : 0;
        __syncthreads();
        ts[tid] = v + u;
        __syncthreads();
    }
    uint32_t run = ts[tid] - s;
    for (int b = lo; b < hi; b++) {
        uint32_t v = a[b];
        a[b] = run;
        run += v;
    }
    if (tid == SRT_T - 1) a[NB] = ts[SRT_T - 1];
}

__global__ __launch_bounds__(SRT_T) void k_srt_scatter(const int16_t* __restrict__ digits, size_t n2, size_t npad, int CH, int NB, int w0,
                                                       const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ bstart,
                                                       uint2* __restrict__ ent) {
    extern __shared__ uint32_t srt_h[];
    const int tid = threadIdx.x, chunk = blockIdx.x, w = w0 + blockIdx.y, nchunks = gridDim.x;
    const uint32_t* o = cnt + ((size_t)w * nchunks + chunk) * NB;
    const uint32_t* bs = bstart + (size_t)w * (NB + 1);
    for (int b = tid; b < NB; b += SRT_T) srt_h[b] = bs[b] + o[b];
    __syncthreads();
    const size_t base = (size_t)chunk * CH;
    if (base >= n2) return;
    const uint4* dg = reinterpret_cast<const uint4*>(digits + (size_t)w * npad + base);
    uint2* E = ent + (size_t)w * npad;
    for (int i = tid; i < CH / 8; i += SRT_T) {
        const uint4 q = dg[i];
#pragma unroll
        for (int h = 0; h < 8; h++) {
            const int d = srt_digit(q, h);
            if (d != 0) {
                const int b = (d < 0 ? -d : d) - 1;
                const uint32_t pos = atomicAdd(&srt_h[b], 1u);
                E[pos] = make_uint2((uint32_t)(base + 8 * i + h) | (d < 0 ? 0x80000000u : 0u), (uint32_t)b);  // one 8-byte store per entry
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Two-pass bucket sort (large MSMs).  The one-pass scatter above keeps one partly filled 64-byte line per (workgroup,
// bucket) alive - 16384 lines per workgroup at c = 15, far more than the L2 holds - so every 8-byte entry leaves the
// chip as a 32-byte write request of its own (profiles/r02_msm_pmc_write.csv: 1.5 TCC_EA0_WRREQ per entry, 0.03 of them
// 64-byte) and the kernel runs at the HBM transaction rate.  Here the bucket index is split b = (p << lo2) | l:
//   pass 1  k_srt2_hist / k_srt2_chunkscan / k_srt2_pscan / k_srt2_scatter: the same counting sort by the PARTITION p
//           only (NP = NB >> lo2 of them, ~4096 entries each for uniform digits): a workgroup keeps NP lines alive and
//           fills each of them within a few hundred stores, so the L2 merges them into whole-line writes;
//   pass 2  works on TILES: a partition of up to SRT2_CAP entries is one tile, a larger one (the short top window of a
//           GLV half-scalar, small or repeated scalars) is cut into ceil(size / SRT2_CAP) tiles (k_srt2_pscan writes the
//           tile -> partition map).  k_srt2_local, one workgroup per tile: a single-tile partition is loaded (coalesced),
//           ranked by l with LDS atomics, placed in an LDS image of the sorted partition and copied out linearly, and the
//           bucket start offsets (bstart) fall out of the same scan.  The tiles of a multi-tile partition first add their
//           per-bucket counts into bcnt (k_srt2_tilecount); k_srt2_local then scans bcnt, reserves a contiguous range
//           per (tile, bucket) with ONE global atomic each and stores the tile's entries there.  Every input - balanced,
//           range-limited, one single bucket - is sorted by whole-chip parallel tiles; the order inside a bucket is
//           arbitrary, which cannot change a group element.
// Output (ent, bstart) is what the one-pass sort produces, up to the order inside a bucket.
#define SRT2_T 512
#define SRT2_PER 8                     // entries per lane in k_srt2_local's register tile
#define SRT2_CAP (SRT2_T * SRT2_PER)   // 4096 entries = 32 KB LDS image
#define SRT2_MAXBINS 1024              // 2^lo2 <= this
#define SRT2_MIN_N2 ((size_t)1 << 15)  // smaller MSMs keep the one-pass sort (fewer launches, everything sits in the L2)

struct Srt2 {
    int lo2, NP, CH, nch, maxtiles;
};
// partitions of <= SRT2_CAP entries on average, at most SRT2_MAXBINS buckets each; runs of >= 16 entries per (chunk, partition)
static Srt2 srt2_params(const SortedPlan& pl, size_t n2) {
    Srt2 s;
    const int bbits = pl.c - 1;
    int hb = 0;
    while (hb < bbits && (n2 >> hb) > SRT2_CAP) hb++;  // 512 partitions at 2^21 entries: 1024 cost the first pass 50 us more than the
                                                       // multi-tile second pass of the larger half of the partitions costs (r3_c.log)
    if (bbits - hb > 10) hb = bbits - 10;
    if (hb > 13) hb = 13;
    if (const char* e = bp_tune("BP_SRT2_HB")) hb = atoi(e) >= 0 && atoi(e) <= 13 && atoi(e) <= bbits && bbits - atoi(e) <= 10 ? atoi(e) : hb;  // TUNING
    s.lo2 = bbits - hb;
    s.NP = 1 << hb;
    size_t ch = (size_t)16 << hb;
    if (ch < 4096) ch = 4096;
    if (ch > (size_t)pl.CH) ch = (size_t)pl.CH;
    s.CH = (int)ch;
    s.nch = (int)(pl.npad / ch);
    s.maxtiles = s.NP + (int)((pl.npad + SRT2_CAP - 1) / SRT2_CAP);
    return s;
}
// 32-bit words of the count workspace: cnt + off ([W][nch][NP] each), pstart ([W][NP + 1]), tmap ([W][maxtiles + 1]:
// tile -> partition, last entry = tiles of the window), tfirst ([W][NP]: first tile of a partition), bcnt + gcur ([W][NB] each)
struct Srt2Ws {
    uint32_t *cnt, *off, *pstart, *tmap, *tfirst, *bcnt, *gcur;
    size_t words;
};
static Srt2Ws srt2_carve(uint32_t* base, const SortedPlan& pl, const Srt2& s) {
    Srt2Ws w;
    const size_t W = (size_t)pl.W;
    w.cnt = base;
    w.off = w.cnt + W * s.nch * s.NP;
    w.pstart = w.off + W * s.nch * s.NP;
    w.tmap = w.pstart + W * (s.NP + 1);
    w.tfirst = w.tmap + W * (s.maxtiles + 1);
    w.bcnt = w.tfirst + W * s.NP;
    w.gcur = w.bcnt + W * pl.NB;
    w.words = (size_t)(w.gcur + W * pl.NB - base);
    return w;
}
size_t srt2_cnt_words(const SortedPlan& pl, size_t n2) { return srt2_carve(nullptr, pl, srt2_params(pl, n2)).words; }

__global__ __launch_bounds__(SRT2_T) void k_srt2_hist(const int16_t* __restrict__ digits, size_t n2, size_t npad, int CH, int NP, int lo2, int w0,
                                                      uint32_t* __restrict__ cnt, uint32_t* __restrict__ bcnt, uint32_t* __restrict__ gcur, int NB) {
    extern __shared__ uint32_t srt_h[];
    const int tid = threadIdx.x, chunk = blockIdx.x, w = w0 + blockIdx.y, nchunks = gridDim.x;
    // this window's slice of the multi-tile counters starts at zero (used by k_srt2_tilecount / k_srt2_local)
    for (int b = chunk * SRT2_T + tid; b < NB; b += nchunks * SRT2_T) bcnt[(size_t)w * NB + b] = gcur[(size_t)w * NB + b] = 0;
    for (int b = tid; b < NP; b += SRT2_T) srt_h[b] = 0;
    __syncthreads();
    const size_t base = (size_t)chunk * CH;
    const uint4* dg = reinterpret_cast<const uint4*>(digits + (size_t)w * npad + base);
    if (base < n2) {
        for (int i = tid; i < CH / 8; i += SRT2_T) {
            const uint4 q = dg[i];
#pragma unroll
            for (int h = 0; h < 8; h++) {
                const int d = srt_digit(q, h);
                if (d != 0) atomicAdd(&srt_h[((d < 0 ? -d : d) - 1) >> lo2], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t* o = cnt + ((size_t)w * nchunks + chunk) * NP;
    for (int b = tid; b < NP; b += SRT2_T) o[b] = srt_h[b];
}

// cnt[w][chunk][p] -> off[w][chunk][p] = entries of partition p in the chunks before `chunk`; ptot[w][p] = entries of p.
// Block (64 partitions, 16 chunk groups): every lane reads its few counts with independent loads.
__global__ __launch_bounds__(1024) void k_srt2_chunkscan(const uint32_t* __restrict__ cnt, int nchunks, int NP, int w0, uint32_t* __restrict__ off,
                                                         uint32_t* __restrict__ ptot) {
    __shared__ uint32_t gsum[16][64];
    const int x = threadIdx.x, y = threadIdx.y, w = w0 + blockIdx.y, p = blockIdx.x * 64 + x;
    const int cpg = (nchunks + 15) / 16;
    int c0 = y * cpg, c1 = c0 + cpg;
    if (c0 > nchunks) c0 = nchunks;
    if (c1 > nchunks) c1 = nchunks;
    const uint32_t* c = cnt + (size_t)w * nchunks * NP;
    uint32_t* o = off + (size_t)w * nchunks * NP;
    uint32_t sum = 0;
    if (p < NP)
        for (int ch = c0; ch < c1; ch++) sum += c[(size_t)ch * NP + p];
    gsum[y][x] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (int yy = 0; yy < y; yy++) run += gsum[yy][x];
    if (p < NP) {
        for (int ch = c0; ch < c1; ch++) {
            const uint32_t v = c[(size_t)ch * NP + p];
            o[(size_t)ch * NP + p] = run;
            run += v;
        }
        if (y == 15) ptot[(size_t)w * (NP + 1) + p] = run;
    }
}

// block-wide inclusive scan of one value per lane (1024 lanes)
__device__ __forceinline__ uint32_t srt2_scan1024(uint32_t v, uint32_t* ts, int tid) {
    ts[tid] = v;
    __syncthreads();
    for (int o2 = 1; o2 < 1024; o2 <<= 1) {
        const uint32_t a = ts[tid];
        const uint32_t u = tid >= o2 ? ts[tid - o2] : 0;
        __syncthreads();
        ts[tid] = a + u;
        __syncthreads();
    }
    return ts[tid];
}
// in place: ptot[w][0 .. NP) -> exclusive starts, entry NP = non-zero digits of the window; tiles of every partition
// (max(1, ceil(size / SRT2_CAP)): an empty partition still has its bucket starts written): tfirst[w][p] = first tile,
// tmap[w][tile] = partition, tmap[w][maxtiles] = tiles of the window.  One workgroup per window.
__global__ __launch_bounds__(1024) void k_srt2_pscan(uint32_t* __restrict__ pstart, int NP, int maxtiles, int w0, uint32_t* __restrict__ tmap,
                                                     uint32_t* __restrict__ tfirst) {
    __shared__ uint32_t ts[1024];
    const int tid = threadIdx.x, w = w0 + blockIdx.x;
    uint32_t* a = pstart + (size_t)w * (NP + 1);
    uint32_t* tm = tmap + (size_t)w * (maxtiles + 1);
    uint32_t* tf = tfirst + (size_t)w * NP;
    const int per = (NP + 1023) / 1024;
    int lo = tid * per, hi = lo + per;
    if (lo > NP) lo = NP;
    if (hi > NP) hi = NP;
    uint32_t s = 0, nt = 0;
    for (int b = lo; b < hi; b++) {
        const uint32_t v = a[b];
        s += v;
        nt += v <= SRT2_CAP ? 1u : (v + SRT2_CAP - 1) / SRT2_CAP;
    }
    const uint32_t incl = srt2_scan1024(s, ts, tid);
    const uint32_t total = ts[1023];
    __syncthreads();
    const uint32_t tincl = srt2_scan1024(nt, ts, tid);
    uint32_t run = incl - s, trun = tincl - nt;
    for (int b = lo; b < hi; b++) {
        const uint32_t v = a[b];
        const uint32_t k = v <= SRT2_CAP ? 1u : (v + SRT2_CAP - 1) / SRT2_CAP;
        a[b] = run;
        tf[b] = trun;
        for (uint32_t t = 0; t < k; t++) tm[trun + t] = (uint32_t)b;
        run += v;
        trun += k;
    }
    if (tid == 1023) {
        a[NP] = total;
        tm[maxtiles] = ts[1023];
    }
}

__global__ __launch_bounds__(SRT2_T) void k_srt2_scatter(const int16_t* __restrict__ digits, size_t n2, size_t npad, int CH, int NP, int lo2, int w0,
                                                         const uint32_t* __restrict__ off, const uint32_t* __restrict__ pstart,
                                                         uint2* __restrict__ tmp) {
    extern __shared__ uint32_t srt_h[];
    const int tid = threadIdx.x, chunk = blockIdx.x, w = w0 + blockIdx.y, nchunks = gridDim.x;
    const size_t base = (size_t)chunk * CH;
    if (base >= n2) return;
    const uint32_t* o = off + ((size_t)w * nchunks + chunk) * NP;
    const uint32_t* ps = pstart + (size_t)w * (NP + 1);
    for (int b = tid; b < NP; b += SRT2_T) srt_h[b] = ps[b] + o[b];
    __syncthreads();
    const uint4* dg = reinterpret_cast<const uint4*>(digits + (size_t)w * npad + base);
    uint2* E = tmp + (size_t)w * npad;
    for (int i = tid; i < CH / 8; i += SRT2_T) {
        const uint4 q = dg[i];
#pragma unroll
        for (int h = 0; h < 8; h++) {
            const int d = srt_digit(q, h);
            if (d != 0) {
                const int b = (d < 0 ? -d : d) - 1;
                const uint32_t pos = atomicAdd(&srt_h[b >> lo2], 1u);
                E[pos] = make_uint2((uint32_t)(base + 8 * i + h) | (d < 0 ? 0x80000000u : 0u), (uint32_t)b);
            }
        }
    }
}

// tile of a workgroup: partition p, entries [s + t * SRT2_CAP, ...) of the window's partition-sorted array
struct Srt2Tile {
    int p;
    uint32_t s, cnt;    // start of the partition, entries of THIS tile
    uint32_t t, ntile;  // tile index inside the partition, tiles of the partition
};
__device__ __forceinline__ bool srt2_tile(const uint32_t* __restrict__ pstart, const uint32_t* __restrict__ tmap, const uint32_t* __restrict__ tfirst,
                                          int NP, int maxtiles, int w, Srt2Tile& T) {
    const uint32_t* tm = tmap + (size_t)w * (maxtiles + 1);
    if (blockIdx.x >= tm[maxtiles]) return false;
    T.p = (int)tm[blockIdx.x];
    const uint32_t* ps = pstart + (size_t)w * (NP + 1);
    T.s = ps[T.p];
    const uint32_t size = ps[T.p + 1] - T.s;
    T.t = blockIdx.x - tfirst[(size_t)w * NP + T.p];
    T.ntile = size <= SRT2_CAP ? 1u : (size + SRT2_CAP - 1) / SRT2_CAP;
    const uint32_t lo = T.t * SRT2_CAP;
    T.cnt = size - lo < SRT2_CAP ? size - lo : SRT2_CAP;
    return true;
}

// multi-tile partitions only: bcnt[w][b] += entries of bucket b in this tile (bcnt zeroed before)
__global__ __launch_bounds__(SRT2_T) void k_srt2_tilecount(const uint2* __restrict__ tmp, const uint32_t* __restrict__ pstart,
                                                           const uint32_t* __restrict__ tmap, const uint32_t* __restrict__ tfirst, int NP, int maxtiles,
                                                           int lo2, int NB, size_t npad, int w0, uint32_t* __restrict__ bcnt) {
    __shared__ uint32_t lh[SRT2_MAXBINS];
    const int tid = threadIdx.x, w = w0 + blockIdx.y;
    Srt2Tile T;
    if (!srt2_tile(pstart, tmap, tfirst, NP, maxtiles, w, T) || T.ntile == 1) return;
    const int bins = 1 << lo2;
    const uint32_t mask = (uint32_t)bins - 1;
    for (int b = tid; b < bins; b += SRT2_T) lh[b] = 0;
    __syncthreads();
    const uint2* in = tmp + (size_t)w * npad + T.s + (size_t)T.t * SRT2_CAP;
    for (uint32_t j = tid; j < T.cnt; j += SRT2_T) atomicAdd(&lh[in[j].y & mask], 1u);
    __syncthreads();
    uint32_t* bc = bcnt + (size_t)w * NB + ((size_t)T.p << lo2);
    for (int b = tid; b < bins; b += SRT2_T)
        if (lh[b]) atomicAdd(&bc[b], lh[b]);
}

__global__ __launch_bounds__(SRT2_T) void k_srt2_local(const uint2* __restrict__ tmp, const uint32_t* __restrict__ pstart,
                                                       const uint32_t* __restrict__ tmap, const uint32_t* __restrict__ tfirst, int NP, int maxtiles, int lo2,
                                                       int NB, size_t npad, int w0, const uint32_t* __restrict__ bcnt, uint32_t* __restrict__ gcur,
                                                       uint2* __restrict__ ent, uint32_t* __restrict__ bstart) {
    __shared__ uint2 img[SRT2_CAP];
    __shared__ uint32_t hs[SRT2_MAXBINS], lh[SRT2_MAXBINS], tsum[SRT2_T];
    const int tid = threadIdx.x, w = w0 + blockIdx.y;
    Srt2Tile T;
    if (!srt2_tile(pstart, tmap, tfirst, NP, maxtiles, w, T)) return;
    const int bins = 1 << lo2;
    const uint32_t mask = (uint32_t)bins - 1;
    const bool multi = T.ntile > 1;
    const uint2* in = tmp + (size_t)w * npad + T.s + (size_t)T.t * SRT2_CAP;
    uint2* out = ent + (size_t)w * npad + T.s;
    uint32_t* bs = bstart + (size_t)w * (NB + 1) + ((size_t)T.p << lo2);
    const uint32_t* bc = bcnt + (size_t)w * NB + ((size_t)T.p << lo2);
    uint32_t* gc = gcur + (size_t)w * NB + ((size_t)T.p << lo2);
    for (int b = tid; b < bins; b += SRT2_T) {
        hs[b] = multi ? bc[b] : 0;
        lh[b] = 0;
    }
    __syncthreads();
    uint2 ev[SRT2_PER];
    uint32_t rk[SRT2_PER];
#pragma unroll
    for (int i = 0; i < SRT2_PER; i++) {
        const uint32_t j = (uint32_t)i * SRT2_T + tid;
        if (j < T.cnt) ev[i] = in[j];
    }
#pragma unroll
    for (int i = 0; i < SRT2_PER; i++) {
        const uint32_t j = (uint32_t)i * SRT2_T + tid;
        if (j < T.cnt) rk[i] = atomicAdd(&lh[ev[i].y & mask], 1u);
    }
    __syncthreads();
    if (!multi)
        for (int b = tid; b < bins; b += SRT2_T) hs[b] = lh[b];
    __syncthreads();
    // exclusive scan of hs[0 .. bins) (bucket sizes of the whole partition): each lane owns `per` consecutive bins
    {
        const int per = (bins + SRT2_T - 1) / SRT2_T;
        int lo = tid * per, hi = lo + per;
        if (lo > bins) lo = bins;
        if (hi > bins) hi = bins;
        uint32_t sum = 0;
        for (int b = lo; b < hi; b++) sum += hs[b];
        tsum[tid] = sum;
        __syncthreads();
        for (int o2 = 1; o2 < SRT2_T; o2 <<= 1) {
            const uint32_t v = tsum[tid];
            const uint32_t u = tid >= o2 ? tsum[tid - o2] : 0;
            __syncthreads();
            tsum[tid] = v + u;
            __syncthreads();
        }
        uint32_t run = tsum[tid] - sum;
        for (int b = lo; b < hi; b++) {
            const uint32_t v = hs[b];
            hs[b] = run;
            if (T.t == 0) bs[b] = T.s + run;
            run += v;
        }
        if (T.t == 0 && T.p == NP - 1 && tid == SRT2_T - 1) bs[bins] = T.s + tsum[SRT2_T - 1];  // bstart[w][NB]
    }
    __syncthreads();
    if (!multi) {
#pragma unroll
        for (int i = 0; i < SRT2_PER; i++) {
            const uint32_t j = (uint32_t)i * SRT2_T + tid;
            if (j < T.cnt) img[hs[ev[i].y & mask] + rk[i]] = ev[i];
        }
        __syncthreads();
        for (uint32_t j = tid; j < T.cnt; j += SRT2_T) out[j] = img[j];
    } else {
        // one contiguous range per (tile, bucket), reserved with one global atomic
        for (int b = tid; b < bins; b += SRT2_T)
            if (lh[b]) hs[b] += atomicAdd(&gc[b], lh[b]);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SRT2_PER; i++) {
            const uint32_t j = (uint32_t)i * SRT2_T + tid;
            if (j < T.cnt) out[hs[ev[i].y & mask] + rk[i]] = ev[i];
        }
    }
}

// The loop is software-pipelined so that a wave does not need its neighbours to hide the gathers (a group of few
// windows runs at one or two waves per SIMD): the point of entry j + 1 is requested before the addition of entry j
// starts.  Entries are read one 64-byte line (8 entries, four 16-byte loads) at a time into an entry-major LDS slot
// (conflict-free 8-byte reads); j0 is a multiple of 8 and the window's array 64-byte aligned.  With one 8-byte global
// load per iteration the line was fetched again for every entry (the point gathers sweep the caches in between):
// TCC_EA read requests 27.5 M per launch against 10.6 M algorithmic.
__global__ __launch_bounds__(256, 2) void k_srt_acc(const affine_mem* __restrict__ pts, const affine_mem* __restrict__ lam, size_t n1,
                                                    const uint2* __restrict__ ent, const uint32_t* __restrict__ bstart, size_t npad, int NB, int L, int nseg,
                                                    int pstride, int w0, int nw, xyzz* __restrict__ Bk, xyzz* __restrict__ Pseg) {
    __shared__ uint2 line[8][256];
    const int tid = threadIdx.x;
    const size_t g = (size_t)blockIdx.x * 256 + tid;
    const int w = w0 + (int)(g / nseg), t = (int)(g % nseg);
    if (w >= w0 + nw) return;
    const uint32_t Nw = bstart[(size_t)w * (NB + 1) + NB];
    const uint32_t j0 = (uint32_t)t * L;
    if (j0 >= Nw) return;
    const uint32_t j1 = (j0 + L < Nw) ? j0 + L : Nw;
    const uint2* E = ent + (size_t)w * npad;
    xyzz acc = xyzz_inf();
    bool atstart = true;
    // line of entries j0 .. j0 + 7 (entries past Nw are never used)
    {
        const uint4* q = reinterpret_cast<const uint4*>(E + j0);
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const uint4 v = q[h];
            line[2 * h][tid] = make_uint2(v.x, v.y);
            line[2 * h + 1][tid] = make_uint2(v.z, v.w);
        }
    }
    uint2 en = line[0][tid];
    int cur = (int)en.y;
    affine_mem nxt;
    {
        const size_t v = en.x & 0x7fffffffu;
        nxt = v < n1 ? pts[v] : lam[v - n1];
    }
    for (uint32_t j = j0; j < j1; j++) {
        const uint2 ec = en;
        affine p = affine_load(nxt);
        if (j + 1 < j1) {
            const uint32_t h = (j + 1 - j0) & 7u;
            if (h == 0) {
                const uint4* q = reinterpret_cast<const uint4*>(E + j + 1);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint4 v = q[k];
                    line[2 * k][tid] = make_uint2(v.x, v.y);
                    line[2 * k + 1][tid] = make_uint2(v.z, v.w);
                }
            }
            en = line[h][tid];
            const size_t v = en.x & 0x7fffffffu;
            nxt = v < n1 ? pts[v] : lam[v - n1];
        }
        const int b = (int)ec.y;
        if (b != cur) {
            if (atstart)
                Pseg[(size_t)w * pstride + t] = acc;
            else
                Bk[(size_t)w * NB + cur] = acc;
            acc = xyzz_inf();
            cur = b;
            atstart = false;
        }
        xyzz_madd_s(acc, p, ec.x >> 31);
    }
    if (atstart)
        Pseg[(size_t)w * pstride + t] = acc;
    else
        Bk[(size_t)w * NB + cur] = acc;
}

// Bucket grid of one window.  A window whose entries all lie in the first NBe = NB / 2^j buckets (always the
// short top window of a GLV half-scalar; any window when the scalars are small) is reduced over an NBe-bucket
// grid b = hi * LOe + lo, so that its few, large buckets spread over all rows and columns instead of a handful.
// Both reduction kernels derive the same grid from bstart (uniform loads).
__device__ __forceinline__ void srt_grid(const uint32_t* __restrict__ bs, int NB, int& lo_bits, int& NBe) {
    const uint32_t Nw = bs[NB];
    int nbits = 31 - __builtin_clz((unsigned)NB);
    while (nbits > 1 && bs[1 << (nbits - 1)] == Nw) nbits--;
    NBe = 1 << nbits;
    if (NBe != NB) lo_bits = (nbits + 1) / 2;
}

// wsum[w] = sum_b (b + 1) B_b = sum_lo lo C_lo + LO sum_hi hi R_hi + sum_hi R_hi from the column sums C and the row sums R.
// Latency-bound: one workgroup per window, 64 QUADS of lanes that share every point operation (ecq.h) - one wave per SIMD,
// so a level costs one operation's latency (with a lane of its own per element the 16 waves of a CU share four issue ports
// and a level costs four).  Quads 0..31 take the columns, 32..63 the rows; a quad owns `per` consecutive elements X_e:
//   local   s_k = sum_j X_(k per + j),  w_k = sum_j j X_(k per + j)           running sums, 2 per - 1 additions
//   scan    suf_k = sum_(k' >= k) s_k'                                          5 levels
//   V_k = w_k + per * suf_k (k >= 1)   ->  sum_k V_k = sum_e e X_e;  rows: V_k <- LO * V_k, and V_0 += suf_0 = sum_hi R_hi
//   tree over the 64 V_k                                                        6 levels
// dbl_c > 0 (one window group, full MSM): the workgroup of window w goes on to 2^(dbl_c w) times its sum, so that the windows'
// doubling chains run side by side - the longest, c (W - 1) doublings, is the Horner chain's length minus one window - and
// k_msm_horner only adds the W results (2^16 points: k_msm_horner 312 us -> the top window's 290 us inside this kernel).
__global__ __launch_bounds__(256) void k_srt_window(const uint32_t* __restrict__ bstart, const xyzz* __restrict__ RC, int NB, int lo_bits, int w0,
                                                       xyzz* __restrict__ wsum, int dbl_c) {
    __shared__ xyzz red[64];
    __builtin_amdgcn_s_setprio(3);  // latency-bound: issue ahead of the co-resident accumulation waves
    const int lq = threadIdx.x >> 2, q = threadIdx.x & 3, w = w0 + blockIdx.x;
    const int stride = (1 << lo_bits) + (NB >> lo_bits);
    int NBe;
    srt_grid(bstart + (size_t)w * (NB + 1), NB, lo_bits, NBe);
    const int LO = 1 << lo_bits, HI = NBe >> lo_bits;
    const int part = lq >> 5, k = lq & 31;
    const int cnt = part ? HI : LO;
    const int per = cnt > 32 ? cnt / 32 : 1;  // LO, HI are powers of two
    const int nq = cnt / per;
    int lp = 0;
    while ((1 << lp) < per) lp++;
    const xyzz* X = RC + (size_t)w * stride + (part ? LO : 0) + (size_t)k * per;
    xyzz run = xyzz_inf(), tot = xyzz_inf();
    if (k < nq) {
        for (int j = per - 1; j >= 1; j--) {
            const xyzz v = X[j];
            xyzzq_add(q, run, v);
            xyzzq_add(q, tot, run);
        }
        const xyzz v = X[0];
        xyzzq_add(q, run, v);
    }
    // suffix scan of s_k inside each part
    if (q == 0) red[lq] = run;
    __syncthreads();
    xyzz suf = run;
    for (int off = 1; off < 32; off <<= 1) {
        const bool has = k < nq && k + off < nq;
        xyzz u;
        if (has) u = red[lq + off];
        __syncthreads();
        if (has) {
            xyzzq_add(q, suf, u);
            if (q == 0) red[lq] = suf;
        }
        __syncthreads();
    }
    xyzz V = tot;
    if (k >= 1 && k < nq) {
        const xyzz t = xyzzq_dbl_n(q, suf, lp);
        xyzzq_add(q, V, t);
    }
    if (part) {
        V = xyzzq_dbl_n(q, V, lo_bits);
        if (k == 0) xyzzq_add(q, V, suf);
    }
    if (q == 0) red[lq] = V;
    __syncthreads();
    for (int off = 32; off >= 1; off >>= 1) {
        if (lq < off) {
            xyzz a = red[lq];
            xyzz u = red[lq + off];
            xyzzq_add(q, a, u);
            if (q == 0) red[lq] = a;
        }
        __syncthreads();
    }
    if (dbl_c > 0 && w > 0) {
        if (threadIdx.x < 4) {  // quad 0
            const xyzz r = xyzzq_dbl_n(q, red[0], dbl_c * w);
            if (q == 0) wsum[w] = r;
        }
    } else if (threadIdx.x == 0) {
        wsum[w] = red[0];
    }
}

// wsum[w_lo] = sum_{w_lo <= w < w_hi} 2^(c (w - w_lo)) wsum[w]  (+ wsum[vidx] when vidx >= 0: the combination of the
// window groups above, already multiplied by 2^(c (w_hi - w_lo)) by k_srt_dbl): Horner over one group's windows, one
// quad (ecq.h), so that this serial chain runs on a side stream while the lower windows are still accumulating.
__global__ __launch_bounds__(64) void k_srt_horner_part(xyzz* __restrict__ wsum, int w_lo, int w_hi, int c, int vidx) {
    if (threadIdx.x >= 4 || blockIdx.x != 0) return;
    __builtin_amdgcn_s_setprio(3);
    const int q = threadIdx.x;
    xyzz acc = wsum[w_hi - 1];
    for (int w = w_hi - 2; w >= w_lo; w--) {
        acc = xyzzq_dbl_n(q, acc, c);
        xyzz o = wsum[w];
        xyzzq_add(q, acc, o);
    }
    if (vidx >= 0) {
        xyzz o = wsum[vidx];
        xyzzq_add(q, acc, o);
    }
    if (q == 0) wsum[w_lo] = acc;
}
// wsum[idx] *= 2^k: the doublings the finished groups' combination needs before the next group's windows can be added
// do not wait for that group's bucket reduction
__global__ __launch_bounds__(64) void k_srt_dbl(xyzz* __restrict__ wsum, int idx, int k) {
    if (threadIdx.x >= 4 || blockIdx.x != 0) return;
    __builtin_amdgcn_s_setprio(3);
    const xyzz r = xyzzq_dbl_n((int)threadIdx.x, wsum[idx], k);
    if (threadIdx.x == 0) wsum[idx] = r;
}

// Row and column sums straight from k_srt_acc's outputs (no per-bucket finalize pass): the sum over a set of
// buckets only needs every partial that belongs to one of them - Bk[b] (the run that starts inside a segment)
// and the Pseg[t] of every segment that starts inside the bucket - in any order.  One workgroup per output:
// o < LO -> C_o = sum_hi bucket[hi*LO + o];  o >= LO -> R_(o-LO) = sum_lo bucket[(o-LO)*LO + lo].
// The workgroup's blockDim.x / 4 QUADS are its logical lanes (ecq.h: a quad shares every point addition, 2.2 times less
// latency than a lane of its own; `tid`, `T` below count quads).  A row's segments are one contiguous range; a column gives
// each of its HI buckets T/HI quads (at least one), and buckets with many segments (the short top window, skewed scalars)
// are summed by the whole workgroup afterwards.  Depth: a few additions per quad + a log2(T)-level LDS tree, whatever the
// bucket sizes are.  LDS: T XYZZ slots, passed as dynamic shared memory.
template <int QW>
__device__ __forceinline__ void rc_add(int q, xyzz& acc, const xyzz& o) {
    if (QW == 4)
        xyzzq_add(q, acc, o);
    else
        xyzz_add(acc, o);
}
template <int QW>  // lanes per logical lane: 4 = a quad shares every addition (lowest latency: the last window group, small
                   // MSMs), 1 = a lane of its own (highest throughput: reductions that run beside the next group's accumulation)
__global__ __launch_bounds__(256) void k_srt_rowcol2(const uint32_t* __restrict__ bstart, const xyzz* __restrict__ Bk,
                                                        const xyzz* __restrict__ Pseg, int NB, int L, int pstride, int lo_bits, int w0,
                                                        xyzz* __restrict__ RC) {
    extern __shared__ __align__(16) unsigned char srt_rc_lds[];
    xyzz* red = reinterpret_cast<xyzz*>(srt_rc_lds);
    __builtin_amdgcn_s_setprio(3);  // latency-bound: issue ahead of the co-resident accumulation waves
    __shared__ uint32_t big_t0[256], big_t1[256];
    __shared__ uint32_t nbig;
    const int tid = threadIdx.x / QW, q = threadIdx.x % QW, T = blockDim.x / QW, w = w0 + blockIdx.y, o = blockIdx.x;  // T = 64 logical lanes per output
    const int stride = (1 << lo_bits) + (NB >> lo_bits);
    const uint32_t* bs = bstart + (size_t)w * (NB + 1);
    int NBe;
    srt_grid(bs, NB, lo_bits, NBe);
    const int LO = 1 << lo_bits, HI = NBe >> lo_bits, nout = LO + HI;
    if (o >= nout) return;
    const xyzz* B = Bk + (size_t)w * NB;
    const xyzz* P = Pseg + (size_t)w * pstride;
    const uint32_t uL = (uint32_t)L;
    xyzz acc = xyzz_inf();
    if (threadIdx.x == 0) nbig = 0;
    __syncthreads();
    if (o >= LO) {
        const int b0 = (o - LO) * LO;
        for (int i = tid; i < LO; i += T) {
            const uint32_t s = bs[b0 + i], e = bs[b0 + i + 1];
            if (e > s && (s % uL)) {
                xyzz v = B[b0 + i];
                rc_add<QW>(q, acc, v);
            }
        }
        const uint32_t t0 = (bs[b0] + uL - 1) / uL, t1 = (bs[b0 + LO] + uL - 1) / uL;
        for (uint32_t t = t0 + tid; t < t1; t += T) {
            xyzz v = P[t];
            rc_add<QW>(q, acc, v);
        }
    } else {
        const int Q = HI >= T ? 1 : T / HI;  // quads per bucket
        for (int h0 = 0; h0 < HI; h0 += T / Q) {
            const int hi = h0 + tid / Q, sub = tid % Q;
            if (hi < HI) {
                const int b = hi * LO + o;
                const uint32_t s = bs[b], e = bs[b + 1];
                if (e > s) {
                    if (sub == 0 && (s % uL)) {
                        xyzz v = B[b];
                        rc_add<QW>(q, acc, v);
                    }
                    const uint32_t t0 = (s + uL - 1) / uL, t1 = (e + uL - 1) / uL;
                    if (t1 - t0 <= (uint32_t)(4 * Q < 8 ? 8 : 4 * Q)) {
                        for (uint32_t t = t0 + sub; t < t1; t += Q) {
                            xyzz v = P[t];
                            rc_add<QW>(q, acc, v);
                        }
                    } else if (sub == 0 && q == 0) {
                        const uint32_t qi = atomicAdd(&nbig, 1u);
                        big_t0[qi] = t0;
                        big_t1[qi] = t1;
                    }
                }
            }
            __syncthreads();
            const uint32_t nq = nbig;
            for (uint32_t qi = 0; qi < nq; qi++) {
                const uint32_t t1 = big_t1[qi];
                for (uint32_t t = big_t0[qi] + tid; t < t1; t += T) {
                    xyzz v = P[t];
                    rc_add<QW>(q, acc, v);
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) nbig = 0;
            __syncthreads();
        }
    }
    if (q == 0) red[tid] = acc;
    __syncthreads();
    for (int off = T / 2; off >= 1; off >>= 1) {
        if (tid < off) {
            xyzz v = red[tid];
            xyzz u = red[tid + off];
            rc_add<QW>(q, v, u);
            if (q == 0) red[tid] = v;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) RC[(size_t)w * stride + o] = red[0];
}

// counting sort by bucket of the windows [w0, w0 + nw) on stream st
static void srt_sort_windows(const int16_t* digits, size_t n2, const SortedPlan& pl, const SortedBufs& b, int w0, int nw, hipStream_t st) {
    const int NB = pl.NB;
    if (b.tmp && (n2 >= SRT2_MIN_N2 || pl.sort2)) {  // two-pass sort (whole-line writes)
        const Srt2 s2 = srt2_params(pl, n2);
        const Srt2Ws ws = srt2_carve(b.cnt, pl, s2);
        dim3 gcw((unsigned)s2.nch, (unsigned)nw), gtw((unsigned)s2.maxtiles, (unsigned)nw);
        const size_t lds = (size_t)s2.NP * 4;
        hipLaunchKernelGGL(k_srt2_hist, gcw, dim3(SRT2_T), lds, st, digits, n2, pl.npad, s2.CH, s2.NP, s2.lo2, w0, ws.cnt, ws.bcnt, ws.gcur, NB);
        hipLaunchKernelGGL(k_srt2_chunkscan, dim3((unsigned)((s2.NP + 63) / 64), (unsigned)nw), dim3(64, 16), 0, st, ws.cnt, s2.nch, s2.NP, w0, ws.off,
                           ws.pstart);
        hipLaunchKernelGGL(k_srt2_pscan, dim3((unsigned)nw), dim3(1024), 0, st, ws.pstart, s2.NP, s2.maxtiles, w0, ws.tmap, ws.tfirst);
        hipLaunchKernelGGL(k_srt2_scatter, gcw, dim3(SRT2_T), lds, st, digits, n2, pl.npad, s2.CH, s2.NP, s2.lo2, w0, ws.off, ws.pstart, b.tmp);
        hipLaunchKernelGGL(k_srt2_tilecount, gtw, dim3(SRT2_T), 0, st, b.tmp, ws.pstart, ws.tmap, ws.tfirst, s2.NP, s2.maxtiles, s2.lo2, NB, pl.npad, w0,
                           ws.bcnt);
        hipLaunchKernelGGL(k_srt2_local, gtw, dim3(SRT2_T), 0, st, b.tmp, ws.pstart, ws.tmap, ws.tfirst, s2.NP, s2.maxtiles, s2.lo2, NB, pl.npad, w0,
                           ws.bcnt, ws.gcur, b.ent, b.bstart);
        return;
    }
    const size_t lds = (size_t)NB * 4;
    dim3 gcw((unsigned)pl.nchunks, (unsigned)nw);
    hipLaunchKernelGGL(k_srt_hist, gcw, dim3(SRT_T), lds, st, digits, n2, pl.npad, pl.CH, NB, w0, b.cnt);
    hipLaunchKernelGGL(k_srt_colscan, dim3((unsigned)((NB + 255) / 256), (unsigned)nw), dim3(256), 0, st, b.cnt, pl.nchunks, NB, w0, b.bstart);
    hipLaunchKernelGGL(k_srt_bstart, dim3((unsigned)nw), dim3(SRT_T), 0, st, b.bstart, NB, w0);
    hipLaunchKernelGGL(k_srt_scatter, gcw, dim3(SRT_T), lds, st, digits, n2, pl.npad, pl.CH, NB, w0, b.cnt, b.bstart, b.ent);
}
// bucket accumulation of the windows [w0, w0 + nw) on stream st, segments of L entries (nseg per window)
static void srt_acc_windows(const affine_mem* pts, const affine_mem* lam, size_t n1, const SortedPlan& pl, const SortedBufs& b, int w0, int nw, int L,
                            int nseg, hipStream_t st) {
    size_t nthr = (size_t)nw * nseg;
    hipLaunchKernelGGL(k_srt_acc, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, pts, lam, n1, b.ent, b.bstart, pl.npad, pl.NB, L, nseg,
                       pl.pstride, w0, nw, b.Bk, b.Pseg);
}
// bucket reduction of the windows [w0, w0 + nw): wsum[w] = sum_b (b+1) bucket_w[b]
static hipError_t srt_reduce_windows(const SortedPlan& pl, const SortedBufs& b, int w0, int nw, int L, xyzz* wsum, hipStream_t st, bool lat,
                                     int dbl_c = 0) {
    const int NB = pl.NB;
    const int nout = (1 << pl.lo_bits) + (NB >> pl.lo_bits);
    // 64 QUADS per output that share every addition (ecq.h), 256-lane workgroups (rowcol2 for one window 142 -> 58 us).  For the
    // groups that reduce beside an accumulation single lanes looked better at first (gpurun_out/r3_j.log, with a high-priority
    // reduction stream); with ordinary streams quads win at every size: 2^19 points 1.51 -> 1.41 ms, 2^20 2.08 -> 2.05, 2^22 equal;
    // 16 or 32 quads per output are slower (2^20: 2.22 ms) - gpurun_out/r3_mid_msm2.log.
    (void)lat;
    int T = 64;
    if (const char* e = bp_tune("BP_MSM_RC_T")) {  // TUNING: "<last group's T>,<earlier groups' T>", 0 = single lanes
        int a = atoi(e), b2 = a;
        if (const char* cpos = strchr(e, ',')) b2 = atoi(cpos + 1);
        T = lat ? a : b2;
    }
    if (T == 16 || T == 32 || T == 64)
        hipLaunchKernelGGL(k_srt_rowcol2<4>, dim3((unsigned)nout, (unsigned)nw), dim3((unsigned)(4 * T)), (size_t)T * sizeof(xyzz), st, b.bstart, b.Bk, b.Pseg,
                           NB, L, pl.pstride, pl.lo_bits, w0, b.RC);
    else
        hipLaunchKernelGGL(k_srt_rowcol2<1>, dim3((unsigned)nout, (unsigned)nw), dim3(64), (size_t)64 * sizeof(xyzz), st, b.bstart, b.Bk, b.Pseg, NB, L,
                           pl.pstride, pl.lo_bits, w0, b.RC);
    hipLaunchKernelGGL(k_srt_window, dim3((unsigned)nw), dim3(256), 0, st, b.bstart, b.RC, NB, pl.lo_bits, w0, wsum, dbl_c);
    return hipGetLastError();
}

// Window sums of the sorted path.
// ng <= 1: every window accumulated and reduced on st; wsum[0 .. W) are the window sums, *W_eff = W, *c_eff = c.
// ng >= 2 (needs ss): the windows are processed top-down in groups of gsz[0], gsz[1], ... windows.  The groups
// accumulate back to back on st.  Side streams: st2 sorts the later groups' windows beside the first accumulation and
// then runs the Horner chain (st4 is an alias of st2: the process has four hardware queues and a third side stream
// shared st3's), st3 runs each finished group's bucket reduction.  Horner chain: V <- 2^(c gsz[g]) V is queued as soon
// as V (the combination of the groups above g) exists, the group's own chain sum_w 2^(c (w - lo)) S_w + V follows its
// reduction.  A last group of ONE window is not combined here: wsum[0] = S_0, wsum[1] = the doubled V, *W_eff = 2,
// *c_eff = 0 - the caller's final kernel adds them on its way to the affine result.
// Otherwise on return (stream order of st) wsum[0] is the complete combination and *W_eff = 1.
hipError_t launch_msm_sorted(const affine_mem* pts, const affine_mem* lam, size_t n1, const int16_t* digits, size_t n2, const SortedPlan& pl,
                             const SortedBufs& b, xyzz* wsum, hipStream_t st, const SortedStreams* ss, const int* gsz, int ng, int* W_eff,
                             int* c_eff) {
    const int W = pl.W;
    hipError_t e;
    *c_eff = pl.c;
    if (!ss || ng <= 1) {
        srt_sort_windows(digits, n2, pl, b, 0, W, st);
        srt_acc_windows(pts, lam, n1, pl, b, 0, W, pl.L, pl.nseg, st);
        *W_eff = W;
        if (pl.weighted) *c_eff = 0;  // the window sums come back multiplied by their 2^(c w): the caller only adds them
        return srt_reduce_windows(pl, b, 0, W, pl.L, wsum, st, true, pl.weighted ? pl.c : 0);
    }
    // Segment length per group: an accumulation launch should fill the chip's resident slots about once (k_srt_acc: three
    // 256-lane workgroups per CU) - a launch of 1.67 "rounds" runs its last round two thirds empty, a launch of a third of a
    // round (one window at the plan's L) is latency-bound at one wave per SIMD: 360 us instead of 160 for the last window
    // at 2^20 (gpurun_out/r3_f.log).  L_g = group entries / resident lanes, a multiple of 8, at least pl.Lmin.
    size_t glanes[SortedStreams::kMaxGroups];
    for (int g = 0; g < SortedStreams::kMaxGroups; g++) glanes[g] = (size_t)(pl.ncu > 0 ? pl.ncu : 256) * (g == ng - 1 ? 256 : g == 0 ? 512 : 384);
    if (const char* e2 = bp_tune("BP_MSM_LANES")) {  // TUNING: "a,b,c" lanes per group, the last value repeats
        size_t v = 0;
        for (int g = 0; g < SortedStreams::kMaxGroups; g++) {
            if (*e2) {
                v = (size_t)atol(e2);
                while (*e2 && *e2 != ',') e2++;
                if (*e2 == ',') e2++;
            }
            if (v >= 4096) glanes[g] = v;
        }
    }
    // Sorts stay on st (a sort beside an accumulation ran 5-20 times slower than alone and cost the accumulation a third,
    // profiles/r02_msm_timeline.txt): the top group's windows first, the other windows right behind the top group's
    // accumulation - there the memory-bound sort runs beside the top group's latency-bound bucket reduction (st3), which
    // leaves most of the chip idle.  BP_MSM_SORTALL=1 (tuning): every window up front.
    bool sortall = true;
    if (const char* e3 = bp_tune("BP_MSM_SORTALL")) sortall = atoi(e3) != 0;  // TUNING
    if (sortall)
        srt_sort_windows(digits, n2, pl, b, 0, W, st);
    else
        srt_sort_windows(digits, n2, pl, b, W - gsz[0], gsz[0], st);
    int hi = W, vidx = -1;
    *W_eff = 1;
    for (int g = 0; g < ng; g++) {
        const int lo = hi - gsz[g];
        int L = pl.L;
        {
            const size_t lanes = glanes[g];
            size_t l = ((size_t)gsz[g] * n2 + lanes - 1) / lanes;
            l = (l + 7) & ~(size_t)7;
            if (l > 256) l = 256;
            if (l < (size_t)pl.Lmin) l = (size_t)pl.Lmin;
            L = (int)l;
        }
        const int nseg = (int)((pl.npad + L - 1) / L);
        srt_acc_windows(pts, lam, n1, pl, b, lo, hi - lo, L, nseg, st);
        if ((e = hipEventRecord(ss->ev_acc[g], st)) != hipSuccess) return e;
        if (g == 0 && !sortall && ng > 1) srt_sort_windows(digits, n2, pl, b, 0, W - gsz[0], st);
        if ((e = hipStreamWaitEvent(ss->st3, ss->ev_acc[g], 0)) != hipSuccess) return e;
        if ((e = srt_reduce_windows(pl, b, lo, hi - lo, L, wsum, ss->st3, g == ng - 1)) != hipSuccess) return e;
        if ((e = hipEventRecord(ss->ev_red[g], ss->st3)) != hipSuccess) return e;
        if (vidx >= 0) hipLaunchKernelGGL(k_srt_dbl, dim3(1), dim3(64), 0, ss->st4, wsum, vidx, pl.c * gsz[g]);
        if ((e = hipStreamWaitEvent(ss->st4, ss->ev_red[g], 0)) != hipSuccess) return e;
        if (g == ng - 1 && gsz[g] == 1 && vidx == 1) {
            *W_eff = 2;
            *c_eff = 0;
        } else {
            hipLaunchKernelGGL(k_srt_horner_part, dim3(1), dim3(64), 0, ss->st4, wsum, lo, hi, pl.c, vidx);
            vidx = lo;
        }
        hi = lo;
    }
    if ((e = hipEventRecord(ss->ev_join, ss->st4)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(st, ss->ev_join, 0)) != hipSuccess) return e;
    return hipGetLastError();
}

}  // namespace bp

BP_OPCOUNT_READER(bp_opcount_msm_sorted)
