// Scalar side of the batched `RangeProof::aggregated_verify`
// (/root/reference/src/proofs/range_proof.rs:565-740), split into short kernels whose launch shape
// matches the parallelism each step really has (the reference does all of it on one thread):
//
//   k_bpv_hash1  4 lanes/proof   y = H(A,S), x = H(T1,T2,G,H), x_u = H(tau_x,miu,tx), c = H(A,S,T1,T2,G,H)
//   k_bpv_fixmul 4 lanes/proof   y*G and x_u*G (:571, :588) from the generator table, 2 lanes each
//   k_bpv_fixinv 1 lane/proof    one shared field inversion -> affine yG, ux (needed as hash input)
//   k_bpv_hash2  k+1 lanes/proof z = H(yG), x_j = H(L_j, R_j, ux)                         (:572, :656)
//   k_bpv_consts 1 lane/proof    one batched inversion mod q for y, y-1, x_j; delta; scalars of
//                                G, H, S, L_j, R_j, V_j, T1, T2                          (:639-740)
//   k_bpv_vec    1 lane/(proof,i) g_i and h_i scalars (:682-705) -> signed window digits
//
// SHA-256 messages are assembled byte-wise in LDS (per-lane slots) and compressed from registers.
#include <hip/hip_runtime.h>

#include "ec.h"
#include "ecmul.h"
#include "fq.h"
#include "kernels.h"
#include "ecwave.h"
#include "sha256.h"

namespace bp {

// ------------------------------------------------------------------------------------------------
// LDS-staged SHA-256 of a short message (<= 440 bytes incl. the 4-byte try-and-increment counter).
#define MSG_MAX 448
#define MSG_STRIDE 452  // odd number of words per lane slot: conflict-free lane-strided access

__device__ __forceinline__ void msg_put(uint8_t* buf, int& pos, uint8_t b) { buf[pos++] = b; }
__device__ __forceinline__ void msg_put_words_be(uint8_t* buf, int& pos, const fp_words& w) {
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        uint32_t v = w.w[i];
        buf[pos] = (uint8_t)(v >> 24);
        buf[pos + 1] = (uint8_t)(v >> 16);
        buf[pos + 2] = (uint8_t)(v >> 8);
        buf[pos + 3] = (uint8_t)v;
        pos += 4;
    }
}
// chain_points: 0x04 || X || Y  (curv `to_bytes(false)`)
__device__ __forceinline__ void msg_put_point(uint8_t* buf, int& pos, const affine_mem& p) {
    msg_put(buf, pos, 4);
    msg_put_words_be(buf, pos, p.x);
    msg_put_words_be(buf, pos, p.y);
}
// chain_bigint: minimal big-endian magnitude (zero -> one 0x00 byte)
__device__ __forceinline__ void msg_put_bigint32(uint8_t* buf, int& pos, const uint8_t* be32) {
    int i = 0;
    while (i < 31 && be32[i] == 0) i++;
    for (; i < 32; i++) buf[pos++] = be32[i];
}

__device__ __forceinline__ void sha256_compress_words(uint32_t h[8], uint32_t w[16]) {
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        uint32_t wi;
        if (i < 16) {
            wi = w[i];
        } else {
            uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
            uint32_t s0 = sha256_rotr(w15, 7) ^ sha256_rotr(w15, 18) ^ (w15 >> 3);
            uint32_t s1 = sha256_rotr(w2, 17) ^ sha256_rotr(w2, 19) ^ (w2 >> 10);
            wi = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
            w[i & 15] = wi;
        }
        uint32_t S1 = sha256_rotr(e, 6) ^ sha256_rotr(e, 11) ^ sha256_rotr(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = hh + S1 + ch + SHA256_K_DEV[i] + wi;
        uint32_t S0 = sha256_rotr(a, 2) ^ sha256_rotr(a, 13) ^ sha256_rotr(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        hh = g;
        g = f;
        f = e;
        e = d + t1;
        d = c;
        c = b;
        b = a;
        a = t1 + t2;
    }
    h[0] += a;
    h[1] += b;
    h[2] += c;
    h[3] += d;
    h[4] += e;
    h[5] += f;
    h[6] += g;
    h[7] += hh;
}

// result_scalar (curv DigestExt, try-and-increment): H(msg || be32(i)), i = 0, 1, ..; first digest < q.
// `len` = message length without the counter.  buf must have room for len + 4 + 73 bytes.
__device__ __forceinline__ fq msg_result_scalar(uint8_t* buf, int len) {
    for (uint32_t ctr = 0;; ctr++) {
        int pos = len;
        buf[pos++] = (uint8_t)(ctr >> 24);
        buf[pos++] = (uint8_t)(ctr >> 16);
        buf[pos++] = (uint8_t)(ctr >> 8);
        buf[pos++] = (uint8_t)ctr;
        const uint32_t bits = (uint32_t)pos * 8u;
        buf[pos++] = 0x80;
        while ((pos & 63) != 56) buf[pos++] = 0;
        buf[pos++] = 0;
        buf[pos++] = 0;
        buf[pos++] = 0;
        buf[pos++] = 0;
        buf[pos++] = (uint8_t)(bits >> 24);
        buf[pos++] = (uint8_t)(bits >> 16);
        buf[pos++] = (uint8_t)(bits >> 8);
        buf[pos++] = (uint8_t)bits;
        uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
        const int nblk = pos >> 6;
        for (int blk = 0; blk < nblk; blk++) {
            uint32_t w[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const uint8_t* q = buf + blk * 64 + i * 4;
                w[i] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
            }
            sha256_compress_words(h, w);
        }
        fq r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = h[7 - i];
        uint32_t t[8];
        if (fq_sub_q(t, r.v) == 1) return r;  // digest < q
    }
}

// proof byte layout (oracle/encoding.py): A S T1 T2 | tau_x miu tx | L[k] | R[k] | a b
#define PF_SCALARS (4 * 33)
#define PF_AB(k) (4 * 33 + 96 + 66 * (k))

// per-proof challenge slots (plain fq)
#define CH_Y 0
#define CH_X 1
#define CH_XU 2
#define CH_C 3
#define CH_Z 4
#define CH_XJ 5  // k entries
#define CH_COUNT(k) (5 + (k))

// ------------------------------------------------------------------------------------------------
// one of the four transcript hashes of proof pr (job 0..3) -> chal[pr][job]; a non-canonical proof scalar raises *badflag
__device__ __forceinline__ void bpv_hash1_job(size_t pr, int job, uint8_t* buf, const uint8_t* __restrict__ proofs, size_t proof_len, int k, int m,
                                              int nm, const affine_mem* __restrict__ pts, const affine_mem* __restrict__ gens, fq* chal, int* badflag) {
    const int NPT = 4 + 2 * k + m;
    const affine_mem* P = pts + pr * NPT;
    const uint8_t* pf = proofs + pr * proof_len;
    int pos = 0;
    if (job == 0) {  // y = H(A, S)                                     range_proof.rs:565-567
        msg_put_point(buf, pos, P[0]);
        msg_put_point(buf, pos, P[1]);
    } else if (job == 1) {  // x = H(T1, T2, G, H)                      :576-578
        msg_put_point(buf, pos, P[2]);
        msg_put_point(buf, pos, P[3]);
        msg_put_point(buf, pos, gens[2 * nm]);
        msg_put_point(buf, pos, gens[2 * nm + 1]);
    } else if (job == 2) {  // x_u = H(tau_x, miu, tx)                  :581-585
        const uint8_t* sc = pf + PF_SCALARS;
        const uint8_t* ab = pf + PF_AB(k);
        // scalars in the proof must be canonical (< q), as curv's deserialisation yields
        bool okc = fq_be_is_canonical(sc) && fq_be_is_canonical(sc + 32) && fq_be_is_canonical(sc + 64) && fq_be_is_canonical(ab) &&
                   fq_be_is_canonical(ab + 32);
        if (!okc) atomicOr(badflag, 1);
        msg_put_bigint32(buf, pos, sc);
        msg_put_bigint32(buf, pos, sc + 32);
        msg_put_bigint32(buf, pos, sc + 64);
    } else {  // c = H(A, S, T1, T2, G, H)                              :591-593
        msg_put_point(buf, pos, P[0]);
        msg_put_point(buf, pos, P[1]);
        msg_put_point(buf, pos, P[2]);
        msg_put_point(buf, pos, P[3]);
        msg_put_point(buf, pos, gens[2 * nm]);
        msg_put_point(buf, pos, gens[2 * nm + 1]);
    }
    chal[pr * CH_COUNT(k) + job] = msg_result_scalar(buf, pos);
}
__global__ __launch_bounds__(64) void k_bpv_hash1(const uint8_t* __restrict__ proofs, size_t proof_len, int k, int m, int nm, size_t nproofs,
                                                  const affine_mem* __restrict__ pts, const affine_mem* __restrict__ gens, fq* __restrict__ chal,
                                                  int* __restrict__ bad) {
    __shared__ __align__(4) uint8_t lds[64 * MSG_STRIDE];
    size_t gid = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (gid >= nproofs * 4) return;
    size_t pr = gid >> 2;
    if (bad[pr]) return;
    bpv_hash1_job(pr, (int)(gid & 3), lds + threadIdx.x * MSG_STRIDE, proofs, proof_len, k, m, nm, pts, gens, chal, &bad[pr]);
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int window_digit(const fq& kk, int c, int w, uint32_t& carry) {
    const uint32_t half = 1u << (c - 1), mask = (1u << c) - 1;
    int bit = w * c;
    uint32_t v = 0;
    if (bit < 256) {
        v = fq_bits32(kk.v, bit) & mask;
    }
    v += carry;
    if (v > half) {
        carry = 1;
        return (int)v - (int)(1u << c);
    }
    carry = 0;
    return (int)v;
}

__device__ __forceinline__ void store_digits(const fq& kk, int c, int W, tdig_t* __restrict__ out) {
    uint32_t carry = 0;
    for (int w = 0; w < W; w++) out[w] = (tdig_t)window_digit(kk, c, w, carry);
}

// y*G and x_u*G per proof: table gathers + mixed adds.  FM_SUB lanes per multiplication (windows
// dealt round-robin), 2*FM_SUB lanes per proof.  Two lanes each: with 16-bit windows a multiplication is
// only 17 table additions, and every extra tree level costs a full XYZZ addition in all lanes
// (8 lanes per multiplication spent 138M VALU instructions per 16384 proofs, mostly in the tree).
#define FM_SUB 2
__global__ __launch_bounds__(256) void k_bpv_fixmul(int k, int nm, size_t nproofs, const affine_mem* __restrict__ table, int c, int W,
                                                    const fq* __restrict__ chal, xyzz* __restrict__ fixres, const int* __restrict__ bad, int nmul,
                                                    int ibase) {
    const int tid = threadIdx.x, lane = tid % (2 * FM_SUB), sub = lane % FM_SUB, which = lane / FM_SUB;
    const size_t pr = (size_t)blockIdx.x * (256 / (2 * FM_SUB)) + tid / (2 * FM_SUB);
    const int NB = 1 << (c - 1);
    xyzz acc = xyzz_inf();
    if (pr < nproofs && !bad[pr] && which < nmul) {
        fq s = chal[pr * CH_COUNT(k) + (which ? CH_XU : CH_Y)];
        // y multiplies the secp256k1 generator (Point::generator(), range_proof.rs:570-571; table row ibase), x_u the
        // caller's G (:588; row 2nm) - the same row whenever G is the generator
        const size_t pG = which ? (size_t)2 * nm : (size_t)ibase;
        uint32_t carry = 0;
        for (int w = 0; w < W; w++) {
            int d = window_digit(s, c, w, carry);
            if ((w % FM_SUB) == sub && d != 0) {
                affine e = affine_load(table[(pG * W + w) * NB + ((d < 0 ? -d : d) - 1)]);
                xyzz_madd_s(acc, e, d < 0);
            }
        }
    }
    xyzz_wave_tree<FM_SUB>(acc, sub);
    if (sub == 0 && pr < nproofs) fixres[pr * 2 + which] = acc;
}

// affine yG, ux with ONE inversion (Montgomery's trick on the two ZZZ).  y = 0 or x_u = 0 makes a
// result the identity: the reference would panic (:569 mod_inv unwrap) / cannot hash it -> malformed.
__device__ __forceinline__ bool bpv_fixinv_body(const xyzz& a, const xyzz& b, affine_mem* out2) {
    if (xyzz_is_inf(a) || xyzz_is_inf(b)) return false;
    fp inv = fp_inv_var(fp_mul(a.ZZZ, b.ZZZ));
    fp ia = fp_mul(inv, b.ZZZ), ib = fp_mul(inv, a.ZZZ);
    affine r;
    fp zi = fp_mul(a.ZZ, ia);
    r.x = fp_mul(a.X, fp_sqr(zi));
    r.y = fp_mul(a.Y, ia);
    out2[0] = affine_store(r);
    zi = fp_mul(b.ZZ, ib);
    r.x = fp_mul(b.X, fp_sqr(zi));
    r.y = fp_mul(b.Y, ib);
    out2[1] = affine_store(r);
    return true;
}
__global__ __launch_bounds__(64) void k_bpv_fixinv(size_t nproofs, const xyzz* __restrict__ fixres, affine_mem* __restrict__ fixaff,
                                                   int* __restrict__ bad) {
    size_t pr = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (pr >= nproofs || bad[pr]) return;
    if (!bpv_fixinv_body(fixres[pr * 2], fixres[pr * 2 + 1], fixaff + pr * 2)) atomicOr(&bad[pr], 1);
}

// z = H(yG) (job 0) and x_j = H(L_j, R_j, ux) (job 1 + j)
__device__ __forceinline__ void bpv_hash2_job(size_t pr, int job, uint8_t* buf, int k, int m, const affine_mem* __restrict__ pts,
                                              const affine_mem* fixaff, fq* chal) {
    const int NPT = 4 + 2 * k + m;
    const affine_mem* P = pts + pr * NPT;
    int pos = 0;
    if (job == 0) {
        msg_put_point(buf, pos, fixaff[pr * 2]);
        chal[pr * CH_COUNT(k) + CH_Z] = msg_result_scalar(buf, pos);
    } else {
        int j = job - 1;
        msg_put_point(buf, pos, P[4 + j]);
        msg_put_point(buf, pos, P[4 + k + j]);
        msg_put_point(buf, pos, fixaff[pr * 2 + 1]);
        chal[pr * CH_COUNT(k) + CH_XJ + j] = msg_result_scalar(buf, pos);
    }
}
__global__ __launch_bounds__(64) void k_bpv_hash2(int k, int m, size_t nproofs, const affine_mem* __restrict__ pts,
                                                  const affine_mem* __restrict__ fixaff, fq* __restrict__ chal, const int* __restrict__ bad) {
    __shared__ __align__(4) uint8_t lds[64 * MSG_STRIDE];
    size_t gid = (size_t)blockIdx.x * 64 + threadIdx.x;
    const int jobs = k + 1;
    if (gid >= nproofs * jobs) return;
    size_t pr = gid / jobs;
    if (bad[pr]) return;
    bpv_hash2_job(pr, (int)(gid % jobs), lds + threadIdx.x * MSG_STRIDE, k, m, pts, fixaff, chal);
}

// ------------------------------------------------------------------------------------------------
// per-proof constants for k_bpv_vec (fqm: reduced-radix Montgomery domain, fq.h)
#define CS_Z 0
#define CS_A 1
#define CS_B 2
#define CS_XJ 3                        // k : x_j
#define CS_XINV(k) (3 + (k))           // k : x_j^-1
#define CS_YINVP(k) (3 + 2 * (k))      // k : y^-(2^t)
#define CS_ZPOW(k) (3 + 3 * (k))       // m : z^(2+j) R (times R once more: k_bpv_vec multiplies it by a PLAIN 2^r)
// split-index tables for k_bpv_vec: i = hi * 2^LO + lo with LO = k/2 low bits and k - LO high bits;
// s_i = SLO[lo] * SHI[hi], s_i^-1 = s_{nm-1-i} = SLO[~lo] * SHI[~hi], y^-i = YLO[lo] * YHI[hi]
#define CS_LOBITS(k) ((k) / 2)
#define CS_HIBITS(k) ((k) - (k) / 2)
#define CS_SLO(k, m) (3 + 3 * (k) + (m))
#define CS_SHI(k, m) (CS_SLO(k, m) + (1 << CS_LOBITS(k)))
#define CS_YLO(k, m) (CS_SHI(k, m) + (1 << CS_HIBITS(k)))
#define CS_YHI(k, m) (CS_YLO(k, m) + (1 << CS_LOBITS(k)))
#define CS_COUNT(k, m) (CS_YHI(k, m) + (1 << CS_HIBITS(k)))
#define BPV_MAXK 16

// split-index tables of k_bpv_vec from x_j, x_j^-1 and y^-(2^t) (already in cs): one product per entry
// (flipping bit t of the index from x^-1 to x multiplies by x^2)
__device__ __forceinline__ void bpv_split_tables(fqm* cs, int k, int m) {
    const fqm ONE = fqm_one();
    {
        const int LO = CS_LOBITS(k), HI = CS_HIBITS(k);
        fqm *slo = cs + CS_SLO(k, m), *shi = cs + CS_SHI(k, m), *ylo = cs + CS_YLO(k, m), *yhi = cs + CS_YHI(k, m);
        fqm p = ONE;
        for (int t = 0; t < LO; t++) p = fqm_mul(p, cs[CS_XINV(k) + (k - 1 - t)]);
        slo[0] = p;
        ylo[0] = ONE;
        for (int j = 1; j < (1 << LO); j++) {
            const int t = 31 - __builtin_clz((unsigned)j);  // top set bit
            const fqm xm = cs[CS_XJ + (k - 1 - t)];
            slo[j] = fqm_mul(slo[j - (1 << t)], fqm_mul(xm, xm));
            ylo[j] = fqm_mul(ylo[j - (1 << t)], cs[CS_YINVP(k) + t]);
        }
        p = ONE;
        for (int t = 0; t < HI; t++) p = fqm_mul(p, cs[CS_XINV(k) + (k - 1 - LO - t)]);
        shi[0] = p;
        yhi[0] = ONE;
        for (int j = 1; j < (1 << HI); j++) {
            const int t = 31 - __builtin_clz((unsigned)j);
            const fqm xm = cs[CS_XJ + (k - 1 - LO - t)];
            shi[j] = fqm_mul(shi[j - (1 << t)], fqm_mul(xm, xm));
            yhi[j] = fqm_mul(yhi[j - (1 << t)], cs[CS_YINVP(k) + LO + t]);
        }
    }
}

// the same tables entry by entry, for callers with lanes to spare (k_bpv_chain): entry e of the 2 (2^LO + 2^HI) is a direct
// product of its index bits' factors (the values are the same elements of F_q whichever order they are multiplied in)
__device__ __forceinline__ void bpv_split_entry(fqm* cs, int k, int m, int e) {
    const int LO = CS_LOBITS(k), HI = CS_HIBITS(k), nlo = 1 << LO, nhi = 1 << HI;
    const bool ytab = e >= nlo + nhi;
    if (ytab) e -= nlo + nhi;
    const bool hi = e >= nlo;
    const int j = hi ? e - nlo : e, bits = hi ? HI : LO, t0 = hi ? LO : 0;
    fqm p = fqm_one();
    if (ytab) {
        for (int t = 0; t < bits; t++)
            if ((j >> t) & 1) p = fqm_mul(p, cs[CS_YINVP(k) + t0 + t]);
        cs[(hi ? CS_YHI(k, m) : CS_YLO(k, m)) + j] = p;
    } else {
        for (int t = 0; t < bits; t++) p = fqm_mul(p, cs[(((j >> t) & 1) ? CS_XJ : CS_XINV(k)) + (k - 1 - t0 - t)]);
        cs[(hi ? CS_SHI(k, m) : CS_SLO(k, m)) + j] = p;
    }
}

template <bool SPLIT = true>
__device__ __forceinline__ void bpv_consts_body(size_t pr, const uint8_t* __restrict__ proofs, size_t proof_len, int n_bits, int m, int k,
                                                size_t nproofs, const fq* __restrict__ chal, int c, int W, tdig_t* __restrict__ digF,
                                                fqm* __restrict__ scF, fq* __restrict__ varS, fqm* __restrict__ cst, int* __restrict__ bad) {
    if (pr >= nproofs || bad[pr]) return;
    const int nm = n_bits * m;
    const int NV = 3 + 2 * k + m;
    const uint8_t* pf = proofs + pr * proof_len;
    const uint8_t* sc = pf + PF_SCALARS;
    const uint8_t* ab = pf + PF_AB(k);
    const fq* ch = chal + pr * CH_COUNT(k);
    fqm* cs = cst + pr * CS_COUNT(k, m);
    fq* vs = varS + pr * (size_t)NV;
    const fqm ONE = fqm_one();

    fq y = ch[CH_Y];
    fqm y_m = fqm_from_fq(y), z_m = fqm_from_fq(ch[CH_Z]), x_m = fqm_from_fq(ch[CH_X]), xu_m = fqm_from_fq(ch[CH_XU]), c_m = fqm_from_fq(ch[CH_C]);
    fqm taux_m = fqm_from_fq(fq_from_be_reduce(sc)), miu_m = fqm_from_fq(fq_from_be_reduce(sc + 32)), tx_m = fqm_from_fq(fq_from_be_reduce(sc + 64));
    fqm a_m = fqm_from_fq(fq_from_be_reduce(ab)), b_m = fqm_from_fq(fq_from_be_reduce(ab + 32));

    // batched inversion of (y, y - 1, x_0 .. x_{k-1}); a zero x_j / y is `.invert().unwrap()` panicking (:658, :569)
    fqm ym1 = fqm_sub(y_m, ONE);
    const bool y_is_one = fqm_is_zero(ym1);
    if (y_is_one) ym1 = ONE;
    fqm pref[BPV_MAXK + 2];
    fqm run = y_m;
    pref[0] = run;
    run = fqm_mul(run, ym1);
    pref[1] = run;
    bool zero_ch = fq_is_zero(y);
    for (int j = 0; j < k; j++) {
        fq xv = ch[CH_XJ + j];
        zero_ch = zero_ch || fq_is_zero(xv);
        fqm xm = fqm_from_fq(xv);
        cs[CS_XJ + j] = xm;
        run = fqm_mul(run, xm);
        pref[2 + j] = run;
    }
    if (zero_ch) {
        atomicOr(&bad[pr], 1);
        return;
    }
    fqm inv = fqm_inv_var(run);  // division steps: ~6x fewer instructions than the exponentiation (one lane per proof, latency-bound)
    for (int j = k - 1; j >= 0; j--) {
        fqm xm = cs[CS_XJ + j];
        fqm xi = fqm_mul(inv, pref[1 + j]);
        inv = fqm_mul(inv, xm);
        cs[CS_XINV(k) + j] = xi;
        vs[1 + j] = fqm_to_fq(fqm_neg(fqm_mul(xm, xm)));      // L_j : -x_j^2     :665, 730
        vs[1 + k + j] = fqm_to_fq(fqm_neg(fqm_mul(xi, xi)));  // R_j : -x_j^-2    :666, 731
    }
    fqm ym1inv = fqm_mul(inv, pref[0]);
    fqm yinv_m = fqm_mul(inv, ym1);
    // y^-(2^t), y^nm
    fqm yp = yinv_m, ynm = y_m;
    for (int t = 0; t < k; t++) {
        cs[CS_YINVP(k) + t] = yp;
        yp = fqm_mul(yp, yp);
        ynm = fqm_mul(ynm, ynm);
    }
    // sum_{i<nm} y^i = (y^nm - 1)/(y - 1)   (nm if y == 1)                    :606-609
    fqm sum_y;
    if (y_is_one) {
        fq t = fq_zero();
        t.v[0] = (uint32_t)nm;
        sum_y = fqm_from_fq(t);
    } else {
        sum_y = fqm_mul(fqm_sub(ynm, ONE), ym1inv);
    }
    // z^(2+j); sum_d = (2^n - 1) sum_j z^(2+j)                                 :597-599, 621-636
    fqm zsq = fqm_mul(z_m, z_m);
    fqm zp = zsq, sum_z = fqm_zero();
    for (int j = 0; j < m; j++) {
        cs[CS_ZPOW(k) + j] = fqm_mul(zp, fqm_r2());
        sum_z = fqm_add(sum_z, zp);
        vs[1 + 2 * k + j] = fqm_to_fq(fqm_mul(c_m, zp));    // V_j : c z^(2+j)  :734-736
        zp = fqm_mul(zp, z_m);
    }
    fq two_n = fq_zero();  // 2^n - 1 as a plain integer (n_bits <= 64 in every config; general for < 256)
    for (int b = 0; b < n_bits; b++) two_n.v[b >> 5] |= 1u << (b & 31);
    fqm sum_d = fqm_mul(fqm_from_fq(two_n), sum_z);
    // delta = (z - z^2) sum_y - z sum_d                                        :639-642
    fqm delta = fqm_sub(fqm_mul(fqm_sub(z_m, zsq), sum_y), fqm_mul(sum_d, z_m));
    // G : x_u (ab - tx) + c (delta - tx) ; H : miu - c tau_x                   :708-723
    fqm abm = fqm_mul(a_m, b_m);
    fqm sG = fqm_add(fqm_mul(xu_m, fqm_sub(abm, tx_m)), fqm_mul(c_m, fqm_sub(delta, tx_m)));
    fqm sH = fqm_sub(miu_m, fqm_mul(c_m, taux_m));
    if (scF) {  // random-linear-combination mode (rlc_kernels.hip): the scalars themselves, Montgomery form
        scF[pr * (size_t)(2 * nm + 2) + 2 * nm] = sG;
        scF[pr * (size_t)(2 * nm + 2) + 2 * nm + 1] = sH;
    } else {
        tdig_t* dg = digF + pr * (size_t)(2 * nm + 2) * W;
        store_digits(fqm_to_fq(sG), c, W, dg + (size_t)(2 * nm) * W);
        store_digits(fqm_to_fq(sH), c, W, dg + (size_t)(2 * nm + 1) * W);
    }
    // S : -x ; T1 : c x ; T2 : c x^2                                           :727, 739-740
    vs[0] = fqm_to_fq(fqm_neg(x_m));
    fqm cx = fqm_mul(c_m, x_m);
    vs[1 + 2 * k + m] = fqm_to_fq(cx);
    vs[2 + 2 * k + m] = fqm_to_fq(fqm_mul(cx, x_m));
    cs[CS_Z] = z_m;
    cs[CS_A] = a_m;
    cs[CS_B] = b_m;
    if (SPLIT) bpv_split_tables(cs, k, m);
}

__global__ __launch_bounds__(64) void k_bpv_consts(const uint8_t* __restrict__ proofs, size_t proof_len, int n_bits, int m, int k, size_t nproofs,
                                                   const fq* __restrict__ chal, int c, int W, tdig_t* __restrict__ digF, fqm* __restrict__ scF,
                                                   fq* __restrict__ varS, fqm* __restrict__ cst, int* __restrict__ bad) {
    bpv_consts_body((size_t)blockIdx.x * 64 + threadIdx.x, proofs, proof_len, n_bits, m, k, nproofs, chal, c, W, digF, scF, varS, cst, bad);
}

// One lane per (proof, i): g_i : a s_i + z ; h_i : y^-i (b s_i^-1 - d_i) - z          :670-705
// s_i = prod_t (bit_t(i) ? x_{k-1-t} : x_{k-1-t}^-1), s_i^-1 = s_{nm-1-i}; d_i = z^(2+i/n) 2^(i mod n).
// Bulletproofs+ (wip != 0, range_proof_wip.rs:537-557) with the constants of k_wipv_consts:
// G_i : (e r') s_i y^-i + e^2 z ; H_i : (e s') s_i^-1 - e^2 z - (e^2 y^nm z^(2(1+i/n))) 2^(i mod n) y^-i.
__device__ __forceinline__ void bpv_vec_elem(size_t pr, int i, int n_bits, int m, int k, const fqm* cst, int c, int W, tdig_t* __restrict__ digF,
                                             fqm* __restrict__ scF, int wip) {
    const int nm = n_bits * m;
    const fqm* cs = cst + pr * CS_COUNT(k, m);
    const int LO = CS_LOBITS(k), lom = (1 << LO) - 1, him = (1 << CS_HIBITS(k)) - 1;
    const int lo = i & lom, hi = i >> LO;
    const fqm *slo = cs + CS_SLO(k, m), *shi = cs + CS_SHI(k, m);
    fqm s = fqm_mul(slo[lo], shi[hi]);
    fqm sinv = fqm_mul(slo[lom - lo], shi[him - hi]);
    fqm yip = fqm_mul(cs[CS_YLO(k, m) + lo], cs[CS_YHI(k, m) + hi]);
    fqm z_m = cs[CS_Z];
    // 2^r as PLAIN 29-bit limbs times the pre-scaled constant (z-power times R, Montgomery form): one multiplication
    fqm two_r = fqm_zero();
    int r = i % n_bits;
    two_r.v[r / 29] = 1u << (r % 29);
    fqm d_i = fqm_mul(two_r, cs[CS_ZPOW(k) + i / n_bits]);
    fqm gs, hs;
    if (wip) {
        gs = fqm_add(fqm_mul(fqm_mul(cs[CS_A], s), yip), z_m);
        hs = fqm_sub(fqm_sub(fqm_mul(cs[CS_B], sinv), z_m), fqm_mul(d_i, yip));
    } else {
        gs = fqm_add(fqm_mul(cs[CS_A], s), z_m);
        hs = fqm_sub(fqm_mul(yip, fqm_sub(fqm_mul(cs[CS_B], sinv), d_i)), z_m);
    }
    if (scF) {
        scF[pr * (size_t)(2 * nm + 2) + i] = gs;
        scF[pr * (size_t)(2 * nm + 2) + nm + i] = hs;
        return;
    }
    tdig_t* dg = digF + pr * (size_t)(2 * nm + 2) * W;
    store_digits(fqm_to_fq(gs), c, W, dg + (size_t)i * W);
    store_digits(fqm_to_fq(hs), c, W, dg + (size_t)(nm + i) * W);
}
__global__ __launch_bounds__(256) void k_bpv_vec(int n_bits, int m, int k, size_t nproofs, const fqm* __restrict__ cst, int c, int W,
                                                 tdig_t* __restrict__ digF, fqm* __restrict__ scF, const int* __restrict__ bad, int wip) {
    const int nm = n_bits * m;
    size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= nproofs * (size_t)nm) return;
    size_t pr = gid / nm;
    if (bad[pr]) return;
    bpv_vec_elem(pr, (int)(gid % nm), n_bits, m, k, cst, c, W, digF, scF, wip);
}

// ------------------------------------------------------------------------------------------------
size_t bpv_chal_count(int k) { return CH_COUNT(k); }
// in units of sizeof(fq): the constants are stored as fqm (9 words)
size_t bpv_cst_count(int k, int m) { return ((size_t)CS_COUNT(k, m) * sizeof(fqm) + sizeof(fq) - 1) / sizeof(fq); }

// the constants kernel + the small-batch ladders' window tables in one launch (defined with the Bulletproofs+ kernels below)
template <int WIP>
__global__ void k_bpv_consts_pre(unsigned cblocks, const uint8_t* __restrict__ proofs, size_t proof_len, int n_bits, int m, int k, size_t nproofs,
                                 const fq* __restrict__ chal, int c, int W, tdig_t* __restrict__ digF, fqm* __restrict__ scF, fq* __restrict__ varS,
                                 fqm* __restrict__ cst, int* __restrict__ bad, const affine_mem* __restrict__ pts, xyzz* __restrict__ pretab, int lpp, int nparts);
// Small batches, the whole scalar side of a proof in ONE workgroup (64 lanes) of ONE launch: hash1 -> y*G, x_u*G -> affine ->
// hash2 -> constants -> g_i / h_i digits, each step on as many lanes as it has independent work, LDS barriers in between
// (the steps of one proof depend on each other, the proofs do not, so nothing crosses workgroups).  As six launches the
// chain cost 181 us of mostly launch gaps before the constants kernel could start (profiles/r02_small_batch.txt A); the
// workgroups nproofs.. of the same launch build the ladders' window tables (the role k_bpv_consts_pre gave them), which now
// overlap the whole chain instead of the constants alone.  Buffers written and read in here (chal, fixaff, cst) are plain
// pointers: their readers must see the workgroup's own earlier stores.
__global__ __launch_bounds__(64) void k_bpv_chain(const uint8_t* __restrict__ proofs, size_t proof_len, int n_bits, int m, int k, size_t nproofs,
                                                  const affine_mem* __restrict__ pts, const affine_mem* __restrict__ gens,
                                                  const affine_mem* __restrict__ table, int c, int W, tdig_t* __restrict__ digF, fq* __restrict__ varS,
                                                  fq* chal, fqm* cst, affine_mem* fixaff, int* bad, int ibase, int nm_tab, xyzz* __restrict__ pretab,
                                                  int lpp, int nparts) {
    const int tid = threadIdx.x;
    if (blockIdx.x >= nproofs) {  // window tables of the (point, slice) lanes of k_bpv_small: scalar-independent
        const size_t g = (size_t)(blockIdx.x - nproofs) * 64 + tid;
        const size_t pr = g / (size_t)lpp;
        const int idx = (int)(g % (size_t)lpp);
        const int NPT = 4 + 2 * k + m, NV = 3 + 2 * k + m;
        const int nsl = 2 * nparts;
        if (pr >= nproofs || idx >= nsl * NV) return;
        const int v = idx / nsl, slice = idx % nsl;
        const int pi = (v == 0) ? 1 : (v < 1 + 2 * k + m) ? (3 + v) : (v == 1 + 2 * k + m ? 2 : 3);
        const affine P = affine_load(pts[pr * NPT + pi]);
        if (affine_is_inf(P)) return;
        ec_slice_table(P, slice, pretab + (pr * (size_t)lpp + idx) * ECMUL_NT, nparts);
        return;
    }
    __shared__ __align__(4) uint8_t lds[(BPV_MAXK + 1) * MSG_STRIDE];
    __shared__ xyzz red[2];
    __shared__ int s_bad;
    const size_t pr = blockIdx.x;
    if (tid == 0) s_bad = bad[pr];
    __syncthreads();
    if (s_bad) return;
    // y, x, x_u, c
    if (tid < 4) bpv_hash1_job(pr, tid, lds + tid * MSG_STRIDE, proofs, proof_len, k, m, nm_tab, pts, gens, chal, &s_bad);
    __syncthreads();
    if (s_bad) {
        if (tid == 0) atomicOr(&bad[pr], 1);
        return;
    }
    // y * generator (lanes 0..7) and x_u * G (lanes 8..15): windows dealt round-robin, three tree levels
    if (tid < 16) {
        const int which = tid >> 3, sub = tid & 7;
        const int NB = 1 << (c - 1);
        const fq sc = chal[pr * CH_COUNT(k) + (which ? CH_XU : CH_Y)];
        const size_t pG = which ? (size_t)2 * nm_tab : (size_t)ibase;
        xyzz acc = xyzz_inf();
        uint32_t carry = 0;
        for (int w = 0; w < W; w++) {
            const int d = window_digit(sc, c, w, carry);
            if ((w & 7) == sub && d != 0) {
                affine e = affine_load(table[(pG * W + w) * NB + ((d < 0 ? -d : d) - 1)]);
                xyzz_madd_s(acc, e, d < 0);
            }
        }
        xyzz_wave_tree<8>(acc, sub);  // lanes 0..15 = one DPP row
        if (sub == 0) red[which] = acc;
    }
    __syncthreads();
    if (tid == 0 && !bpv_fixinv_body(red[0], red[1], fixaff + pr * 2)) s_bad = 1;
    __syncthreads();
    if (s_bad) {
        if (tid == 0) atomicOr(&bad[pr], 1);
        return;
    }
    // z, x_j
    if (tid < k + 1) bpv_hash2_job(pr, tid, lds + tid * MSG_STRIDE, k, m, pts, fixaff, chal);
    __syncthreads();
    if (tid == 0) bpv_consts_body<false>(pr, proofs, proof_len, n_bits, m, k, nproofs, chal, c, W, digF, nullptr, varS, cst, bad);
    __syncthreads();
    for (int e = tid; e < 2 * ((1 << CS_LOBITS(k)) + (1 << CS_HIBITS(k))); e += 64) bpv_split_entry(cst + pr * CS_COUNT(k, m), k, m, e);
    __syncthreads();
    const int nm = n_bits * m;
    for (int i = tid; i < nm; i += 64) bpv_vec_elem(pr, i, n_bits, m, k, cst, c, W, digF, nullptr, 0);
}

hipError_t launch_bpv_scalar_pipeline(const uint8_t* proofs, size_t proof_len, int n_bits, int m, int k, size_t nproofs, const affine_mem* pts,
                                      const affine_mem* gens, const affine_mem* table, int c, int W, tdig_t* digF, fqm* scF, fq* varS, fq* chal,
                                      fq* cst, xyzz* fixres, affine_mem* fixaff, int* bad, hipStream_t st, int ibase, int nm_tab, xyzz* pretab,
                                      int lpp, int nparts) {
    const int nm = n_bits * m;
    static const bool unchained = bp_tune("BP_VERIFY_SMALL_UNCHAINED") != nullptr;  // A/B: the six-launch form
    // small batch: the whole chain and the window tables in one launch, while chain and table workgroups together stay at one wave
    // per SIMD (n=64 m=1, one call alone: 512 proofs 0.99 -> 0.82 ms, 256 proofs 1.00 -> 0.79; 1024 proofs 1.17 -> 1.42)
    if (pretab && lpp > 0 && !scF && k <= BPV_MAXK && !unchained && nproofs * (size_t)lpp <= 32768) {
        const unsigned blocks = (unsigned)(nproofs + (nproofs * (size_t)lpp + 63) / 64);
        hipLaunchKernelGGL(k_bpv_chain, dim3(blocks), dim3(64), 0, st, proofs, proof_len, n_bits, m, k, nproofs, pts, gens, table, c, W, digF, varS, chal,
                           reinterpret_cast<fqm*>(cst), fixaff, bad, ibase, nm_tab, pretab, lpp, nparts);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_bpv_hash1, dim3((unsigned)((nproofs * 4 + 63) / 64)), dim3(64), 0, st, proofs, proof_len, k, m, nm_tab, nproofs, pts, gens,
                       chal, bad);  // nm_tab: G, H sit after the table's nm_tab-long vectors (>= nm: the proof uses their prefixes)
    hipLaunchKernelGGL(k_bpv_fixmul, dim3((unsigned)((nproofs + 256 / (2 * FM_SUB) - 1) / (256 / (2 * FM_SUB)))), dim3(256), 0, st, k, nm_tab, nproofs, table, c, W, chal, fixres, bad, 2, ibase);
    hipLaunchKernelGGL(k_bpv_fixinv, dim3((unsigned)((nproofs + 63) / 64)), dim3(64), 0, st, nproofs, fixres, fixaff, bad);
    hipLaunchKernelGGL(k_bpv_hash2, dim3((unsigned)((nproofs * (k + 1) + 63) / 64)), dim3(64), 0, st, k, m, nproofs, pts, fixaff, chal, bad);
    const unsigned cblocks = (unsigned)((nproofs + 63) / 64);
    if (pretab && lpp > 0)  // + the ladders' window tables in the same launch (k_bpv_consts_pre)
        hipLaunchKernelGGL((k_bpv_consts_pre<0>), dim3(cblocks + (unsigned)((nproofs * (size_t)lpp + 63) / 64)), dim3(64), 0, st, cblocks, proofs,
                           proof_len, n_bits, m, k, nproofs, chal, c, W, digF, scF, varS, reinterpret_cast<fqm*>(cst), bad, pts, pretab, lpp, nparts);
    else
        hipLaunchKernelGGL(k_bpv_consts, dim3(cblocks), dim3(64), 0, st, proofs, proof_len, n_bits, m, k, nproofs, chal, c, W, digF, scF, varS,
                           reinterpret_cast<fqm*>(cst), bad);
    hipLaunchKernelGGL(k_bpv_vec, dim3((unsigned)((nproofs * (size_t)nm + 255) / 256)), dim3(256), 0, st, n_bits, m, k, nproofs,
                       reinterpret_cast<const fqm*>(cst), c, W, digF, scF, bad, 0);
    return hipGetLastError();
}

// ================================================================================================
// Bulletproofs+ : scalar side of the batched `RangeProofWIP::aggregated_verify`
// (/root/reference/src/proofs/range_proof_wip.rs:419-574).  Same kernel shapes as above:
//   k_wipv_hash1  k+2 lanes/proof  y = H(A), e = H(a_tag, b_tag, g, h), x_j = H(L_j, R_j, g, h)   (:420, :465-467, :477)
//   k_bpv_fixmul  (one multiplication per proof) y*G (:423);  k_wipv_fixinv -> affine
//   k_wipv_hash2  1 lane/proof     z = H(A, yG)                                                 (:424)
//   k_wipv_consts 1 lane/proof     one batched inversion for y, y-1, x_j; zeta; scalars of g, h, A, L_j, R_j, a_tag, V_j
//   k_bpv_vec     (wip = 1)        G_i, H_i scalars -> signed window digits
// Point order in `pts` (k_wipv_decompress): b_tag, A, L[k], R[k], a_tag, V[m]; the variable terms of the MSM are
// pts[1 ..] in that order and the sum is compared with pts[0] = b_tag (:604).
// proof byte layout (oracle/encoding.py): A | L[k] | R[k] | a_tag b_tag | r' s' delta'
#define PFW_SCALARS(k) (33 * (2 * (k) + 3))
#define CH_E CH_X

__device__ __forceinline__ void wipv_hash1_job(size_t pr, int job, uint8_t* buf, const uint8_t* __restrict__ proofs, size_t proof_len, int k, int m,
                                               int nm, const affine_mem* __restrict__ pts, const affine_mem* __restrict__ gens, fq* chal, int* badflag) {
    const int NPT = 3 + 2 * k + m;
    const affine_mem* P = pts + pr * NPT;
    int pos = 0;
    if (job == 0) {  // y = H(A)                                               range_proof_wip.rs:420
        const uint8_t* sc = proofs + pr * proof_len + PFW_SCALARS(k);
        // scalars in the proof must be canonical (< q), as curv's deserialisation yields
        if (!(fq_be_is_canonical(sc) && fq_be_is_canonical(sc + 32) && fq_be_is_canonical(sc + 64))) atomicOr(badflag, 1);
        msg_put_point(buf, pos, P[1]);
        chal[pr * CH_COUNT(k) + CH_Y] = msg_result_scalar(buf, pos);
    } else if (job == 1) {  // e = H(a_tag, b_tag, g, h)                        :465-467
        msg_put_point(buf, pos, P[2 + 2 * k]);
        msg_put_point(buf, pos, P[0]);
        msg_put_point(buf, pos, gens[2 * nm]);
        msg_put_point(buf, pos, gens[2 * nm + 1]);
        chal[pr * CH_COUNT(k) + CH_E] = msg_result_scalar(buf, pos);
    } else {  // x_j = H(L_j, R_j, g, h)                                        :477
        const int j = job - 2;
        msg_put_point(buf, pos, P[2 + j]);
        msg_put_point(buf, pos, P[2 + k + j]);
        msg_put_point(buf, pos, gens[2 * nm]);
        msg_put_point(buf, pos, gens[2 * nm + 1]);
        chal[pr * CH_COUNT(k) + CH_XJ + j] = msg_result_scalar(buf, pos);
    }
}
__global__ __launch_bounds__(64) void k_wipv_hash1(const uint8_t* __restrict__ proofs, size_t proof_len, int k, int m, int nm, size_t nproofs,
                                                   const affine_mem* __restrict__ pts, const affine_mem* __restrict__ gens, fq* __restrict__ chal,
                                                   int* __restrict__ bad) {
    __shared__ __align__(4) uint8_t lds[64 * MSG_STRIDE];
    const int jobs = k + 2;
    size_t gid = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (gid >= nproofs * jobs) return;
    size_t pr = gid / jobs;
    if (bad[pr]) return;
    wipv_hash1_job(pr, (int)(gid % jobs), lds + threadIdx.x * MSG_STRIDE, proofs, proof_len, k, m, nm, pts, gens, chal, &bad[pr]);
}

// affine yG; y = 0 makes it the identity (the reference panics at :458 mod_inv unwrap) -> malformed
__global__ __launch_bounds__(64) void k_wipv_fixinv(size_t nproofs, const xyzz* __restrict__ fixres, affine_mem* __restrict__ fixaff,
                                                    int* __restrict__ bad) {
    size_t pr = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (pr >= nproofs || bad[pr]) return;
    xyzz a = fixres[pr * 2];
    if (xyzz_is_inf(a)) {
        atomicOr(&bad[pr], 1);
        return;
    }
    fixaff[pr * 2] = affine_store(xyzz_to_affine_var(a));
}

// z = H(A, yG)                                                                 :424
__global__ __launch_bounds__(64) void k_wipv_hash2(int k, int m, size_t nproofs, const affine_mem* __restrict__ pts,
                                                   const affine_mem* __restrict__ fixaff, fq* __restrict__ chal, const int* __restrict__ bad) {
    __shared__ __align__(4) uint8_t lds[64 * MSG_STRIDE];
    size_t pr = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (pr >= nproofs || bad[pr]) return;
    uint8_t* buf = lds + threadIdx.x * MSG_STRIDE;
    const int NPT = 3 + 2 * k + m;
    int pos = 0;
    msg_put_point(buf, pos, pts[pr * NPT + 1]);
    msg_put_point(buf, pos, fixaff[pr * 2]);
    chal[pr * CH_COUNT(k) + CH_Z] = msg_result_scalar(buf, pos);
}

template <bool SPLIT = true>
__device__ __forceinline__ void wipv_consts_body(size_t pr, const uint8_t* __restrict__ proofs, size_t proof_len, int n_bits, int m, int k,
                                                 size_t nproofs, const fq* __restrict__ chal, int c, int W, tdig_t* __restrict__ digF,
                                                 fqm* __restrict__ scF, fq* __restrict__ varS, fqm* __restrict__ cst, int* __restrict__ bad) {
    if (pr >= nproofs || bad[pr]) return;
    const int nm = n_bits * m;
    const int NV = 2 + 2 * k + m;
    const uint8_t* sc = proofs + pr * proof_len + PFW_SCALARS(k);
    const fq* ch = chal + pr * CH_COUNT(k);
    fqm* cs = cst + pr * CS_COUNT(k, m);
    fq* vs = varS + pr * (size_t)NV;
    const fqm ONE = fqm_one();

    fq y = ch[CH_Y];
    fqm y_m = fqm_from_fq(y), z_m = fqm_from_fq(ch[CH_Z]), e_m = fqm_from_fq(ch[CH_E]);
    fqm r_m = fqm_from_fq(fq_from_be_reduce(sc)), s_m = fqm_from_fq(fq_from_be_reduce(sc + 32)), dl_m = fqm_from_fq(fq_from_be_reduce(sc + 64));
    fqm esq = fqm_mul(e_m, e_m);
    fqm mesq = fqm_neg(esq);

    // batched inversion of (y, y - 1, x_0 .. x_{k-1}); a zero x_j / y is `.unwrap()` panicking (:458, :479)
    fqm ym1 = fqm_sub(y_m, ONE);
    const bool y_is_one = fqm_is_zero(ym1);
    if (y_is_one) ym1 = ONE;
    fqm pref[BPV_MAXK + 2];
    fqm run = y_m;
    pref[0] = run;
    run = fqm_mul(run, ym1);
    pref[1] = run;
    bool zero_ch = fq_is_zero(y);
    for (int j = 0; j < k; j++) {
        fq xv = ch[CH_XJ + j];
        zero_ch = zero_ch || fq_is_zero(xv);
        fqm xm = fqm_from_fq(xv);
        cs[CS_XJ + j] = xm;
        run = fqm_mul(run, xm);
        pref[2 + j] = run;
    }
    if (zero_ch) {
        atomicOr(&bad[pr], 1);
        return;
    }
    fqm inv = fqm_inv_var(run);  // division steps: ~6x fewer instructions than the exponentiation (one lane per proof, latency-bound)
    for (int j = k - 1; j >= 0; j--) {
        fqm xm = cs[CS_XJ + j];
        fqm xi = fqm_mul(inv, pref[1 + j]);
        inv = fqm_mul(inv, xm);
        cs[CS_XINV(k) + j] = xi;
        vs[1 + j] = fqm_to_fq(fqm_mul(mesq, fqm_mul(xm, xm)));      // L_j : -e^2 x_j^2     :483-487, 560
        vs[1 + k + j] = fqm_to_fq(fqm_mul(mesq, fqm_mul(xi, xi)));  // R_j : -e^2 x_j^-2    :484-492, 561
    }
    fqm ym1inv = fqm_mul(inv, pref[0]);
    fqm yinv_m = fqm_mul(inv, ym1);
    fqm yp = yinv_m, ynm = y_m;
    for (int t = 0; t < k; t++) {
        cs[CS_YINVP(k) + t] = yp;
        yp = fqm_mul(yp, yp);
        ynm = fqm_mul(ynm, ynm);
    }
    // sum_{i=1..nm} y^i = y (y^nm - 1)/(y - 1)   (nm if y == 1)                :523-531
    fqm sum_y;
    if (y_is_one) {
        fq t = fq_zero();
        t.v[0] = (uint32_t)nm;
        sum_y = fqm_from_fq(t);
    } else {
        sum_y = fqm_mul(y_m, fqm_mul(fqm_sub(ynm, ONE), ym1inv));
    }
    const fqm ynm1 = fqm_mul(ynm, y_m);  // y^(nm+1)                            :521
    const fqm esq_ynm = fqm_mul(esq, ynm), mesq_ynm1 = fqm_mul(mesq, ynm1);
    // z^(2(j+1)); sum_d = (2^n - 1) sum_j z^(2(j+1))                          :441-455
    fqm zsq = fqm_mul(z_m, z_m);
    fqm zp = zsq, sum_z = fqm_zero();
    for (int j = 0; j < m; j++) {
        cs[CS_ZPOW(k) + j] = fqm_mul(fqm_mul(esq_ynm, zp), fqm_r2());                  // e^2 y^nm z^(2(j+1)): factor of d_i y^(nm-i) in H_i
        sum_z = fqm_add(sum_z, zp);
        vs[2 + 2 * k + j] = fqm_to_fq(fqm_mul(mesq_ynm1, zp));      // V_j : -e^2 y^(nm+1) z^(2(j+1))   :568-571
        zp = fqm_mul(zp, zsq);
    }
    fq two_n = fq_zero();
    for (int b = 0; b < n_bits; b++) two_n.v[b >> 5] |= 1u << (b & 31);
    fqm sum_d = fqm_mul(fqm_from_fq(two_n), sum_z);
    // zeta = (z - z^2) sum_y - z y^(nm+1) sum_d                                :522-531
    fqm zeta = fqm_sub(fqm_mul(fqm_sub(z_m, zsq), sum_y), fqm_mul(fqm_mul(z_m, ynm1), sum_d));
    // g : r' s' y - e^2 zeta ; h : delta'                                      :532-535, 598
    fqm sG = fqm_sub(fqm_mul(fqm_mul(r_m, s_m), y_m), fqm_mul(esq, zeta));
    if (scF) {  // random-linear-combination mode: the scalars themselves, Montgomery form
        scF[pr * (size_t)(2 * nm + 2) + 2 * nm] = sG;
        scF[pr * (size_t)(2 * nm + 2) + 2 * nm + 1] = dl_m;
    } else {
        tdig_t* dg = digF + pr * (size_t)(2 * nm + 2) * W;
        store_digits(fqm_to_fq(sG), c, W, dg + (size_t)(2 * nm) * W);
        store_digits(fqm_to_fq(dl_m), c, W, dg + (size_t)(2 * nm + 1) * W);
    }
    vs[0] = fqm_to_fq(mesq);                    // A (range proof) : -e^2     :564-565
    vs[1 + 2 * k] = fqm_to_fq(fqm_neg(e_m));    // a_tag : -e                 :574
    cs[CS_Z] = fqm_mul(esq, z_m);               // e^2 z                      :539
    cs[CS_A] = fqm_mul(e_m, r_m);               // e r'                       :538
    cs[CS_B] = fqm_mul(e_m, s_m);               // e s'                       :548
    if (SPLIT) bpv_split_tables(cs, k, m);
}

__global__ __launch_bounds__(64) void k_wipv_consts(const uint8_t* __restrict__ proofs, size_t proof_len, int n_bits, int m, int k, size_t nproofs,
                                                    const fq* __restrict__ chal, int c, int W, tdig_t* __restrict__ digF, fqm* __restrict__ scF,
                                                    fq* __restrict__ varS, fqm* __restrict__ cst, int* __restrict__ bad) {
    wipv_consts_body((size_t)blockIdx.x * 64 + threadIdx.x, proofs, proof_len, n_bits, m, k, nproofs, chal, c, W, digF, scF, varS, cst, bad);
}

// Small batches (k_bpv_small, verify_kernels.hip): the constants kernel and, IN THE SAME LAUNCH, the scalar-independent part of
// the proof-specific ladders.  One lane per proof leaves the GPU empty for the length of the constants' serial chain
// (0.16 ms); workgroups cblocks.. of this launch build, meanwhile, the window table of every (point, slice) lane of
// k_bpv_small - for the upper slices that includes the 68 doublings of their base - so that only the windows themselves
// are left once the scalars exist (512 proofs: 1.12 -> 0.9 ms per call).  Two roles in one launch rather than two
// streams: a process has four hardware queues and the context's side stream carries the other batch in flight.
template <int WIP>
__global__ __launch_bounds__(64) void k_bpv_consts_pre(unsigned cblocks, const uint8_t* __restrict__ proofs, size_t proof_len, int n_bits, int m, int k,
                                                       size_t nproofs, const fq* __restrict__ chal, int c, int W, tdig_t* __restrict__ digF,
                                                       fqm* __restrict__ scF, fq* __restrict__ varS, fqm* __restrict__ cst, int* __restrict__ bad,
                                                       const affine_mem* __restrict__ pts, xyzz* __restrict__ pretab, int lpp, int nparts) {
    if (blockIdx.x < cblocks) {
        const size_t pr = (size_t)blockIdx.x * 64 + threadIdx.x;
        if (WIP)
            wipv_consts_body(pr, proofs, proof_len, n_bits, m, k, nproofs, chal, c, W, digF, scF, varS, cst, bad);
        else
            bpv_consts_body(pr, proofs, proof_len, n_bits, m, k, nproofs, chal, c, W, digF, scF, varS, cst, bad);
        return;
    }
    const size_t g = (size_t)(blockIdx.x - cblocks) * 64 + threadIdx.x;
    const size_t pr = g / (size_t)lpp;
    const int idx = (int)(g % (size_t)lpp);  // one (point, slice) per lane: lpp >= 4 NV (bpv_slice_lanes)
    const int NPT = (WIP ? 3 : 4) + 2 * k + m, NV = (WIP ? 2 : 3) + 2 * k + m;
    const int nsl = 2 * nparts;
    if (pr >= nproofs || idx >= nsl * NV) return;
    const int v = idx / nsl, slice = idx % nsl;
    // variable point order: S, L_0.., R_0.., V_0.., T1, T2  ->  index into pts (as in k_bpv_msm)
    const int pi = WIP ? 1 + v : (v == 0) ? 1 : (v < 1 + 2 * k + m) ? (3 + v) : (v == 1 + 2 * k + m ? 2 : 3);
    const affine P = affine_load(pts[pr * NPT + pi]);
    if (affine_is_inf(P)) return;  // (a malformed proof's slot: its verdict never reads the table)
    ec_slice_table(P, slice, pretab + (pr * (size_t)lpp + idx) * ECMUL_NT, nparts);
}

// Bulletproofs+ form of k_bpv_chain: the scalar side of one proof in one 64-lane workgroup (hash1 on k + 2 lanes -> y*G on eight
// -> affine -> z -> constants -> split tables -> G_i / H_i digits), the window tables of k_bpv_small's ladders in the workgroups
// nproofs.. of the same launch.  Six launches before (launch_wipv_scalar_pipeline below).
__global__ __launch_bounds__(64) void k_wipv_chain(const uint8_t* __restrict__ proofs, size_t proof_len, int n_bits, int m, int k, size_t nproofs,
                                                   const affine_mem* __restrict__ pts, const affine_mem* __restrict__ gens,
                                                   const affine_mem* __restrict__ table, int c, int W, tdig_t* __restrict__ digF, fq* __restrict__ varS,
                                                   fq* chal, fqm* cst, affine_mem* fixaff, int* bad, int ibase, int nm_tab, xyzz* __restrict__ pretab,
                                                   int lpp, int nparts) {
    const int tid = threadIdx.x;
    if (blockIdx.x >= nproofs) {  // window tables of the (point, slice) lanes of k_bpv_small: scalar-independent
        const size_t g = (size_t)(blockIdx.x - nproofs) * 64 + tid;
        const size_t pr = g / (size_t)lpp;
        const int idx = (int)(g % (size_t)lpp);
        const int NPT = 3 + 2 * k + m, NV = 2 + 2 * k + m;
        const int nsl = 2 * nparts;
        if (pr >= nproofs || idx >= nsl * NV) return;
        const int v = idx / nsl, slice = idx % nsl;
        const affine P = affine_load(pts[pr * NPT + 1 + v]);
        if (affine_is_inf(P)) return;
        ec_slice_table(P, slice, pretab + (pr * (size_t)lpp + idx) * ECMUL_NT, nparts);
        return;
    }
    __shared__ __align__(4) uint8_t lds[(BPV_MAXK + 2) * MSG_STRIDE];
    __shared__ xyzz red[1];
    __shared__ int s_bad;
    const size_t pr = blockIdx.x;
    if (tid == 0) s_bad = bad[pr];
    __syncthreads();
    if (s_bad) return;
    // y, e, x_j
    if (tid < k + 2) wipv_hash1_job(pr, tid, lds + tid * MSG_STRIDE, proofs, proof_len, k, m, nm_tab, pts, gens, chal, &s_bad);
    __syncthreads();
    if (s_bad) {
        if (tid == 0) atomicOr(&bad[pr], 1);
        return;
    }
    // y * generator on eight lanes (windows dealt round-robin), three tree levels                  range_proof_wip.rs:423
    if (tid < 8) {
        const int NB = 1 << (c - 1);
        const fq sc = chal[pr * CH_COUNT(k) + CH_Y];
        xyzz acc = xyzz_inf();
        uint32_t carry = 0;
        for (int w = 0; w < W; w++) {
            const int d = window_digit(sc, c, w, carry);
            if ((w & 7) == tid && d != 0) {
                affine e = affine_load(table[((size_t)ibase * W + w) * NB + ((d < 0 ? -d : d) - 1)]);
                xyzz_madd_s(acc, e, d < 0);
            }
        }
        xyzz_wave_tree<8>(acc, tid);
        if (tid == 0) red[0] = acc;
    }
    __syncthreads();
    if (tid == 0) {
        const xyzz a = red[0];
        if (xyzz_is_inf(a)) {  // y = 0 (the reference panics at :458)
            s_bad = 1;
        } else {
            fixaff[pr * 2] = affine_store(xyzz_to_affine_var(a));
            // z = H(A, yG)                                                       :424
            int pos = 0;
            msg_put_point(lds, pos, pts[pr * (3 + 2 * k + m) + 1]);
            msg_put_point(lds, pos, fixaff[pr * 2]);
            chal[pr * CH_COUNT(k) + CH_Z] = msg_result_scalar(lds, pos);
            wipv_consts_body<false>(pr, proofs, proof_len, n_bits, m, k, nproofs, chal, c, W, digF, nullptr, varS, cst, bad);
        }
    }
    __syncthreads();
    if (s_bad) {
        if (tid == 0) atomicOr(&bad[pr], 1);
        return;
    }
    for (int e = tid; e < 2 * ((1 << CS_LOBITS(k)) + (1 << CS_HIBITS(k))); e += 64) bpv_split_entry(cst + pr * CS_COUNT(k, m), k, m, e);
    __syncthreads();
    const int nm = n_bits * m;
    for (int i = tid; i < nm; i += 64) bpv_vec_elem(pr, i, n_bits, m, k, cst, c, W, digF, nullptr, 1);
}

hipError_t launch_wipv_scalar_pipeline(const uint8_t* proofs, size_t proof_len, int n_bits, int m, int k, size_t nproofs, const affine_mem* pts,
                                       const affine_mem* gens, const affine_mem* table, int c, int W, tdig_t* digF, fqm* scF, fq* varS, fq* chal,
                                       fq* cst, xyzz* fixres, affine_mem* fixaff, int* bad, hipStream_t st, int ibase, int nm_tab, xyzz* pretab,
                                       int lpp, int nparts) {
    const int nm = n_bits * m;
    static const bool unchained = bp_tune("BP_VERIFY_SMALL_UNCHAINED") != nullptr;  // A/B: the six-launch form
    if (pretab && lpp > 0 && !scF && k <= BPV_MAXK && !unchained && nproofs * (size_t)lpp <= 32768) {  // as launch_bpv_scalar_pipeline
        const unsigned blocks = (unsigned)(nproofs + (nproofs * (size_t)lpp + 63) / 64);
        hipLaunchKernelGGL(k_wipv_chain, dim3(blocks), dim3(64), 0, st, proofs, proof_len, n_bits, m, k, nproofs, pts, gens, table, c, W, digF, varS,
                           chal, reinterpret_cast<fqm*>(cst), fixaff, bad, ibase, nm_tab, pretab, lpp, nparts);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_wipv_hash1, dim3((unsigned)((nproofs * (k + 2) + 63) / 64)), dim3(64), 0, st, proofs, proof_len, k, m, nm_tab, nproofs, pts,
                       gens, chal, bad);
    hipLaunchKernelGGL(k_bpv_fixmul, dim3((unsigned)((nproofs + 256 / (2 * FM_SUB) - 1) / (256 / (2 * FM_SUB)))), dim3(256), 0, st, k, nm_tab, nproofs,
                       table, c, W, chal, fixres, bad, 1, ibase);
    hipLaunchKernelGGL(k_wipv_fixinv, dim3((unsigned)((nproofs + 63) / 64)), dim3(64), 0, st, nproofs, fixres, fixaff, bad);
    hipLaunchKernelGGL(k_wipv_hash2, dim3((unsigned)((nproofs + 63) / 64)), dim3(64), 0, st, k, m, nproofs, pts, fixaff, chal, bad);
    const unsigned cblocks = (unsigned)((nproofs + 63) / 64);
    if (pretab && lpp > 0)
        hipLaunchKernelGGL((k_bpv_consts_pre<1>), dim3(cblocks + (unsigned)((nproofs * (size_t)lpp + 63) / 64)), dim3(64), 0, st, cblocks, proofs,
                           proof_len, n_bits, m, k, nproofs, chal, c, W, digF, scF, varS, reinterpret_cast<fqm*>(cst), bad, pts, pretab, lpp, nparts);
    else
        hipLaunchKernelGGL(k_wipv_consts, dim3(cblocks), dim3(64), 0, st, proofs, proof_len, n_bits, m, k, nproofs, chal, c, W, digF, scF, varS,
                           reinterpret_cast<fqm*>(cst), bad);
    hipLaunchKernelGGL(k_bpv_vec, dim3((unsigned)((nproofs * (size_t)nm + 255) / 256)), dim3(256), 0, st, n_bits, m, k, nproofs,
                       reinterpret_cast<const fqm*>(cst), c, W, digF, scF, bad, 1);
    return hipGetLastError();
}

}  // namespace bp
