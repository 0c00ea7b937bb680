// secp256k1 base field F_p, p = 2^256 - 2^32 - 977, for CDNA4's 32-bit VALU.
//
// Replaces the field arithmetic libsecp256k1 performs behind curv-kzen's `Point<Secp256k1>` operators
// (reference call sites: every `&Point * &Scalar` / `Point + Point`, e.g.
// /root/reference/src/proofs/range_proof.rs:773-774).
//
// Representation: 9 limbs of 29 bits ("reduced radix", value = sum v[i] * 2^(29 i)).  Measured on
// MI355X (profiles/r01_peak_mad32.jsonl): v_mad_u64_u32 issues in ~5 cycles per wave, but every
// carry instruction around it (v_addc_co_u32, 64-bit add) costs ~4, so a saturated 8x32 schoolbook
// spends more time on carries than on multiplies.  With 29-bit limbs a whole product column
// (9 products < 2^60) accumulates in one 64-bit register pair with NO carry handling: the product is
// a pure `v_mad_u64_u32 acc, a, b, acc` chain (81 MADs), carries are resolved once per column with a
// shift and a mask.  Reduction uses the pseudo-Mersenne fold 2^256 = 2^32 + 977 (mod p): values stay
// in the plain domain, bit-identical to the canonical coordinates the reference serialises.
//
// Invariant of every fp value at rest ("weakly normalised"):
//     v[0..7] <= 2^29 + 2^12,   v[8] <= 2^24 + 2^12      (value < 2^256 + 2^45, not nec. < p)
// Canonical form (unique, < p) is produced only where bytes leave (fp_to_words) or for equality tests.
//
// Storage format in HBM / at the boundary is the canonical 8 x 32-bit words (`fp_words`, 32 bytes).
// Every function is __host__ __device__: the same code is unit-tested on the CPU (tests/).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BP_HD __host__ __device__ __forceinline__
#define BP_HD_NOINLINE inline __host__ __device__ __noinline__
// fp_mul / fp_sqr are ~200 VALU instructions each.  Inlined they avoid ~30 argument moves and the
// call sequence per multiply (measured with bench_micro/ecops: xyzz_madd 5.5 us -> 4.0 us per wave);
// -DBP_FP_MUL_NOINLINE keeps them as real functions (smaller code, for experiments).
#ifdef BP_FP_MUL_NOINLINE
#define BP_HD_MUL inline __host__ __device__ __noinline__
#else
#define BP_HD_MUL __host__ __device__ __forceinline__
#endif
#else
#define BP_HD inline
#define BP_HD_NOINLINE inline
#define BP_HD_MUL inline
#endif

// Optimisation barrier on a 32-bit value (device only): stops value-range facts from flowing into
// the multiplier when fp_mul is inlined (see DESIGN.md "inlined field multiply").
#if defined(__HIP_DEVICE_COMPILE__)
#define BP_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define BP_OPAQUE(x) ((void)0)
#endif
// Which limbs of a multiplier operand go through the barrier: BP_OPAQUE_MODE 0 = all nine (round 1), 1 = limb 8 only (default),
// 2 = none.  The miscompile needs an operand pair the compiler KNOWS to be 24-bit values - limb 8 is masked with 0xFFFFFF by
// every normalisation - so hiding that one limb is enough: bench_micro/ecops checksums are identical to mode 0 for every
// operation and differ for every point operation in mode 2 (profiles/r02_fp_variants.txt), and the bit-exact GPU parity suite
// passes.  A barrier on a limb that stays live after the multiplication costs a register copy; nine per operand were ~12 % of
// the instructions of a mixed addition.
#ifndef BP_OPAQUE_MODE
#define BP_OPAQUE_MODE 1
#endif
#if BP_OPAQUE_MODE == 0
#define BP_OPAQUE_OPERAND(f)                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 9; i_++) BP_OPAQUE((f).v[i_])
#elif BP_OPAQUE_MODE == 1
#define BP_OPAQUE_OPERAND(f) BP_OPAQUE((f).v[8])
#else
#define BP_OPAQUE_OPERAND(f) ((void)0)
#endif

// Instrumented build only (-DBP_OPCOUNT, `make opcount` -> libbpmsm_opcount.so; scripts/opcount.py): every lane counts the
// field multiplications and squarings it executes, so that bench.py's `frac_executed` is a counted number, not a model.
// The arrays are file-local: a translation unit that wants its count read back instantiates BP_OPCOUNT_READER(name).
#if defined(BP_OPCOUNT) && defined(__HIPCC__)
#define BP_OPCOUNT_SLOTS (1 << 18)
static __device__ unsigned int bp_oc_mul[BP_OPCOUNT_SLOTS], bp_oc_sqr[BP_OPCOUNT_SLOTS];
#if defined(__HIP_DEVICE_COMPILE__)
#define BP_COUNT(arr) atomicAdd(&arr[(blockIdx.x * blockDim.x + threadIdx.x) & (BP_OPCOUNT_SLOTS - 1)], 1u)
#else
#define BP_COUNT(arr) ((void)0)
#endif
#define BP_OPCOUNT_READER(name)                                                                              \
    extern "C" int name(unsigned long long* mul, unsigned long long* sqr, int reset) {                       \
        static unsigned int h[BP_OPCOUNT_SLOTS];                                                             \
        unsigned long long m = 0, q = 0;                                                                     \
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(bp_oc_mul), sizeof h) != hipSuccess) return -3;                \
        for (int i = 0; i < BP_OPCOUNT_SLOTS; i++) m += h[i];                                                \
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(bp_oc_sqr), sizeof h) != hipSuccess) return -3;                \
        for (int i = 0; i < BP_OPCOUNT_SLOTS; i++) q += h[i];                                                \
        if (reset) {                                                                                         \
            for (int i = 0; i < BP_OPCOUNT_SLOTS; i++) h[i] = 0;                                             \
            if (hipMemcpyToSymbol(HIP_SYMBOL(bp_oc_mul), h, sizeof h) != hipSuccess) return -3;              \
            if (hipMemcpyToSymbol(HIP_SYMBOL(bp_oc_sqr), h, sizeof h) != hipSuccess) return -3;              \
        }                                                                                                    \
        *mul = m;                                                                                            \
        *sqr = q;                                                                                            \
        return 0;                                                                                            \
    }
#else
#define BP_COUNT(arr) ((void)0)
#define BP_OPCOUNT_READER(name)
#endif

#if defined(BP_CHECK_BOUNDS) && !defined(__HIPCC__)
#include <stdlib.h>  // abort() of the bounds-checking host build
#endif
namespace bp {

struct fp {
    uint32_t v[9];
};
struct fp_words {  // canonical value < p as 8 little-endian 32-bit words (memory / wire format)
    uint32_t w[8];
};

#if defined(BP_CHECK_BOUNDS) && !defined(__HIPCC__)
#define BP_LIMB(expr) bp_limb_checked((int64_t)(expr))
inline uint32_t bp_limb_checked(int64_t v) {
    if (v < 0 || v > 0xFFFFFFFFll) abort();
    return (uint32_t)v;
}
#else
#define BP_LIMB(expr) ((uint32_t)(expr))
#endif
#define BP_M29 0x1FFFFFFFu
#define BP_P_C0 977u  // 2^256 - p = 2^32 + 977 = 977 + 8 * 2^29

BP_HD fp fp_zero() {
    fp r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = 0;
    return r;
}
BP_HD fp fp_one() {
    fp r = fp_zero();
    r.v[0] = 1;
    return r;
}

// Carry-propagate limbs (< 2^32 each) and fold bits >= 256 back in: restores the invariant.
BP_HD void fp_normalize(uint32_t v[9]) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        v[i + 1] += v[i] >> 29;
        v[i] &= BP_M29;
    }
    uint32_t e = v[8] >> 24;
    v[8] &= 0x00FFFFFFu;
    v[0] += e * BP_P_C0;
    v[1] += e << 3;
    v[1] += v[0] >> 29;
    v[0] &= BP_M29;
}

// Unique representative in [0, p).
BP_HD fp fp_canon(const fp& a) {
    fp r = a;
    fp_normalize(r.v);
    // full carry chain (no fold): limbs < 2^29, v[8] <= 2^24, value < 2p
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.v[i + 1] += r.v[i] >> 29;
        r.v[i] &= BP_M29;
    }
    // s = r + (2^32 + 977); if bit 256 of s is set then r >= p and r - p = s - 2^256
    uint32_t s[9];
    s[0] = r.v[0] + BP_P_C0;
    s[1] = r.v[1] + 8u + (s[0] >> 29);
    s[0] &= BP_M29;
#pragma unroll
    for (int i = 2; i < 9; i++) {
        s[i] = r.v[i] + (s[i - 1] >> 29);
        s[i - 1] &= BP_M29;
    }
    if (s[8] >> 24) {
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = s[i];
        r.v[8] = s[8] & 0x00FFFFFFu;
    }
    return r;
}

// Zero test.  A value is < 2^256 + 2^45 < 2p, so it is 0 mod p only as 0 or p itself, and its low 29 bits are the low 29
// bits of limb 0 whatever slack the other limbs carry (limbs are non-negative): unless those are 0 or p mod 2^29 the value
// is not zero and the canonicalisation (about a hundred instructions) is skipped.  xyzz_madd / xyzz_add ask this once per
// addition (is the addend +-the accumulator?), where the answer is "no" except with probability 2^-28.
#define BP_P_LOW29 0x1FFFFC2Fu
BP_HD bool fp_is_zero(const fp& a) {
    const uint32_t l0 = a.v[0] & BP_M29;
    if (l0 != 0u && l0 != BP_P_LOW29) return false;
    fp c = fp_canon(a);
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= c.v[i];
    return o == 0;
}

BP_HD fp fp_add(const fp& a, const fp& b) {
    fp r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
    fp_normalize(r.v);
    return r;
}

// a - b = a + K - b with K = 4p written with every limb >= 2^30 - 2 (>= any limb of b).
BP_HD fp fp_sub(const fp& a, const fp& b) {
    const uint32_t K[9] = {0x5ffff0bcu, 0x5fffffddu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x03fffffdu};
    fp r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = BP_LIMB((int64_t)a.v[i] + K[i] - b.v[i]);
    fp_normalize(r.v);
    return r;
}
BP_HD fp fp_neg(const fp& a) { return fp_sub(fp_zero(), a); }

// ---- linear combinations with ONE normalisation, and operands that skip it ------------------------------------------
// fp_add / fp_sub normalise after every operation (~30 instructions, as many as the operation itself); the point formulas
// chain several of them, and every instruction costs a multiply's issue slot (profiles/r02_fp_variants.txt).  The helpers
// below form  m_a a - m_b b - m_c c + (K or 2K)  limb by limb and normalise once.  Inputs are normalised values (limbs
// <= 2^29 + 2^13, limb 8 < 2^24).  Bounds: K = 4p has limbs >= 1.5 * 2^30 - 3 (limb 8: 2^26 - 3), 2K = 8p twice that, so
//   with K :  m_b + m_c <= 2  and  m_a <= 4   (m_a 2^29 + 1.5 * 2^30 < 2^32)
//   with 2K:  m_b + m_c <= 5  and  m_a <= 1   (2^29 + 3 * 2^30 < 2^32)
// keep every limb in [0, 2^32); fp_normalize's carries (<= 7) and its single fold of the top (value < 10 p) restore the invariant.
#define BP_K4P {0x5ffff0bcu, 0x5fffffddu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x03fffffdu}
#define BP_K8P {0xbfffe178u, 0xbfffffbau, 0xbffffffau, 0xbffffffau, 0xbffffffau, 0xbffffffau, 0xbffffffau, 0xbffffffau, 0x07fffffau}
// a - b - c
BP_HD fp fp_sub2(const fp& a, const fp& b, const fp& c) {
    const uint32_t K[9] = BP_K4P;
    fp r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = BP_LIMB((int64_t)a.v[i] + K[i] - b.v[i] - c.v[i]);
    fp_normalize(r.v);
    return r;
}
// a - b - 2c   (X3 of the additions: R^2 - PPP - 2Q)
BP_HD fp fp_sub_b_2c(const fp& a, const fp& b, const fp& c) {
    const uint32_t K[9] = BP_K8P;
    fp r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = BP_LIMB((int64_t)a.v[i] + K[i] - b.v[i] - 2 * (int64_t)c.v[i]);
    fp_normalize(r.v);
    return r;
}
// a - 2b
BP_HD fp fp_sub_2b(const fp& a, const fp& b) {
    const uint32_t K[9] = BP_K4P;
    fp r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = BP_LIMB((int64_t)a.v[i] + K[i] - 2 * (int64_t)b.v[i]);
    fp_normalize(r.v);
    return r;
}
// a - 4b
BP_HD fp fp_sub_4b(const fp& a, const fp& b) {
    const uint32_t K[9] = BP_K8P;
    fp r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = BP_LIMB((int64_t)a.v[i] + K[i] - 4 * (int64_t)b.v[i]);
    fp_normalize(r.v);
    return r;
}
// 3a, 4a
BP_HD fp fp_mul3(const fp& a) {
    fp r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = BP_LIMB(3 * (int64_t)a.v[i]);
    fp_normalize(r.v);
    return r;
}
BP_HD fp fp_mul4(const fp& a) {
    fp r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = BP_LIMB(4 * (int64_t)a.v[i]);
    fp_normalize(r.v);
    return r;
}
// LAZY results: not normalised (limbs < 2^31.4, value a multiple of p off), valid ONLY as one operand of fp_mul whose other
// operand is normalised: a column of nine products stays below 9 * 2^31.4 * 2^29.001 < 2^63.6.  Never into fp_sqr, fp_add,
// fp_sub, fp_is_zero or memory.
BP_HD fp fp_sub_lazy(const fp& a, const fp& b) {  // a - b
    const uint32_t K[9] = BP_K4P;
    fp r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = BP_LIMB((int64_t)a.v[i] + K[i] - b.v[i]);
    return r;
}
BP_HD fp fp_2a_sub_lazy(const fp& a, const fp& b) {  // 2a - b
    const uint32_t K[9] = BP_K4P;
    fp r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = BP_LIMB(2 * (int64_t)a.v[i] + K[i] - b.v[i]);
    return r;
}
BP_HD fp fp_dbl_lazy(const fp& a) {  // 2a
    fp r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = BP_LIMB(2 * (int64_t)a.v[i]);
    return r;
}
BP_HD fp fp_cneg_lazy(const fp& a, bool neg) {  // neg ? -a : a, without a branch
    const uint32_t K[9] = BP_K4P;
    fp r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = neg ? BP_LIMB((int64_t)K[i] - a.v[i]) : a.v[i];
    return r;
}

BP_HD fp fp_dbl(const fp& a) { return fp_add(a, a); }
BP_HD bool fp_eq(const fp& a, const fp& b) { return fp_is_zero(fp_sub(a, b)); }

// Tail of a product: o.v[0..8] hold the nine output limbs after one carry pass, c the carry out of
// limb 8 (weight 2^261) and h8 the top high limb (weight 2^261 * 2^(29*8), folded once already so it
// arrives here as 256*h8 at weight 2^261).  Folds everything at or above 2^256 with 2^256 = 2^32 + 977.
BP_HD void fp_fold_top(fp& o, uint64_t c, uint32_t h8) {
    uint64_t E = (uint64_t)(o.v[8] >> 24) + ((c + ((uint64_t)h8 << 8)) << 5);
    o.v[8] &= 0x00FFFFFFu;
    c = (uint64_t)o.v[0] + E * BP_P_C0;
    o.v[0] = (uint32_t)c & BP_M29;
    c >>= 29;
    c += (uint64_t)o.v[1] + (E << 3);
    o.v[1] = (uint32_t)c & BP_M29;
    c >>= 29;
    c += o.v[2];
    o.v[2] = (uint32_t)c & BP_M29;
    c >>= 29;
    o.v[3] += (uint32_t)c;
}

// c + a * b as ONE v_mad_u64_u32 whose addend is the running accumulator.  Written as `c += (uint64_t)a * b` the compiler
// reassociates every column into a chain of its own that starts at zero and adds the carried accumulator with a 64-bit addition
// afterwards (18 v_lshl_add_u64 per multiplication, 4 issue cycles each); BP_MAD_CHAIN keeps the single chain (inline asm cannot
// be reassociated): fewer instructions, but no instruction-level parallelism inside one multiplication.
#if defined(__HIP_DEVICE_COMPILE__) && defined(BP_MAD_CHAIN)
__device__ __forceinline__ uint64_t bp_mad(uint32_t a, uint32_t b, uint64_t c) {
    uint64_t r, sc;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(sc) : "v"(a), "v"(b), "v"(c));
    return r;
}
#elif defined(BP_CHECK_BOUNDS) && !defined(__HIPCC__)
// host build of the sanitizer tests: a column accumulator that wraps or a limb that leaves [0, 2^32) aborts
inline uint64_t bp_mad(uint32_t a, uint32_t b, uint64_t c) {
    uint64_t r;
    if (__builtin_add_overflow(c, (uint64_t)a * b, &r)) abort();
    return r;
}
#else
BP_HD uint64_t bp_mad(uint32_t a, uint32_t b, uint64_t c) { return c + (uint64_t)a * b; }
#endif

// 9x9 product scanning on v_mad_u64_u32.  The HIGH columns 9..16 are accumulated first and split
// into 29-bit limbs H[k] (they must become multiplier operands).  2^261 = 256 * 2^29 + 31264 (mod p),
// so H[k] (weight 2^261 * 2^(29k)) adds 31264*H[k] to column k and 256*H[k] to column k+1: each LOW
// column is then ONE accumulator chain  carry + sum a_i b_(k-i) + 31264 H[k] + 256 H[k-1]  followed
// by a mask and a shift - no separate carry pass over the low half, no 64-bit additions at all.
// 81 + 17 MADs.  (The multiplication by 256 goes through the multiplier on purpose: as a 64-bit
// shift-and-add it costs two instructions and a register pair with a zeroed high half.)
BP_HD_MUL fp fp_mul(fp a, fp b) {
    BP_COUNT(bp_oc_mul);
    BP_OPAQUE_OPERAND(a);
    BP_OPAQUE_OPERAND(b);
    uint32_t k256 = 256u;
    BP_OPAQUE(k256);
    uint32_t H[9];
    uint64_t c = 0;
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) c = bp_mad(a.v[i], b.v[k - i], c);
        H[k - 9] = (uint32_t)c & BP_M29;
        c >>= 29;
    }
    H[8] = (uint32_t)c;
    fp o;
    c = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) c = bp_mad(a.v[i], b.v[k - i], c);
        c = bp_mad(H[k], 31264u, c);
        if (k > 0) c = bp_mad(H[k - 1], k256, c);
        o.v[k] = (uint32_t)c & BP_M29;
        c >>= 29;
    }
    fp_fold_top(o, c, H[8]);
    return o;
}

// Squaring: 45 MADs (off-diagonal terms use a doubled operand) + 17 in the fold.
BP_HD_MUL fp fp_sqr(fp a) {
    BP_COUNT(bp_oc_sqr);
    BP_OPAQUE_OPERAND(a);
    uint32_t k256 = 256u;
    BP_OPAQUE(k256);
    uint32_t d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;
    uint32_t H[9];
    uint64_t c = 0;
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; 2 * i < k; i++) c = bp_mad(d[i], a.v[k - i], c);
        if ((k & 1) == 0) c = bp_mad(a.v[k / 2], a.v[k / 2], c);
        H[k - 9] = (uint32_t)c & BP_M29;
        c >>= 29;
    }
    H[8] = (uint32_t)c;
    fp o;
    c = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) c = bp_mad(d[i], a.v[k - i], c);
        if ((k & 1) == 0) c = bp_mad(a.v[k / 2], a.v[k / 2], c);
        c = bp_mad(H[k], 31264u, c);
        if (k > 0) c = bp_mad(H[k - 1], k256, c);
        o.v[k] = (uint32_t)c & BP_M29;
        c >>= 29;
    }
    fp_fold_top(o, c, H[8]);
    return o;
}

BP_HD fp fp_sqr_n(fp a, int n) {
    for (int i = 0; i < n; i++) a = fp_sqr(a);
    return a;
}

// a^(p-2) by the standard secp256k1 addition chain (255 squarings + 15 multiplications).
BP_HD_NOINLINE fp fp_inv(fp a) {
    fp x2 = fp_mul(fp_sqr(a), a);
    fp x3 = fp_mul(fp_sqr(x2), a);
    fp x6 = fp_mul(fp_sqr_n(x3, 3), x3);
    fp x9 = fp_mul(fp_sqr_n(x6, 3), x3);
    fp x11 = fp_mul(fp_sqr_n(x9, 2), x2);
    fp x22 = fp_mul(fp_sqr_n(x11, 11), x11);
    fp x44 = fp_mul(fp_sqr_n(x22, 22), x22);
    fp x88 = fp_mul(fp_sqr_n(x44, 44), x44);
    fp x176 = fp_mul(fp_sqr_n(x88, 88), x88);
    fp x220 = fp_mul(fp_sqr_n(x176, 44), x44);
    fp x223 = fp_mul(fp_sqr_n(x220, 3), x3);
    fp t = fp_mul(fp_sqr_n(x223, 23), x22);
    t = fp_mul(fp_sqr_n(t, 5), a);
    t = fp_mul(fp_sqr_n(t, 3), x2);
    t = fp_mul(fp_sqr_n(t, 2), a);
    return t;
}

// a^((p+1)/4): a square root of a when a is a quadratic residue (p = 3 mod 4).
BP_HD_NOINLINE fp fp_sqrt_candidate(fp a) {
    fp x2 = fp_mul(fp_sqr(a), a);
    fp x3 = fp_mul(fp_sqr(x2), a);
    fp x6 = fp_mul(fp_sqr_n(x3, 3), x3);
    fp x9 = fp_mul(fp_sqr_n(x6, 3), x3);
    fp x11 = fp_mul(fp_sqr_n(x9, 2), x2);
    fp x22 = fp_mul(fp_sqr_n(x11, 11), x11);
    fp x44 = fp_mul(fp_sqr_n(x22, 22), x22);
    fp x88 = fp_mul(fp_sqr_n(x44, 44), x44);
    fp x176 = fp_mul(fp_sqr_n(x88, 88), x88);
    fp x220 = fp_mul(fp_sqr_n(x176, 44), x44);
    fp x223 = fp_mul(fp_sqr_n(x220, 3), x3);
    fp t = fp_mul(fp_sqr_n(x223, 23), x22);
    t = fp_mul(fp_sqr_n(t, 6), x2);
    t = fp_sqr_n(t, 2);
    return t;
}

// ---- conversions between the compute form and the canonical 8x32 storage form -------------------
BP_HD fp fp_from_words(const fp_words& s) {
    const uint32_t* w = s.w;
    fp r;
    r.v[0] = w[0] & BP_M29;
    r.v[1] = ((w[0] >> 29) | (w[1] << 3)) & BP_M29;
    r.v[2] = ((w[1] >> 26) | (w[2] << 6)) & BP_M29;
    r.v[3] = ((w[2] >> 23) | (w[3] << 9)) & BP_M29;
    r.v[4] = ((w[3] >> 20) | (w[4] << 12)) & BP_M29;
    r.v[5] = ((w[4] >> 17) | (w[5] << 15)) & BP_M29;
    r.v[6] = ((w[5] >> 14) | (w[6] << 18)) & BP_M29;
    r.v[7] = ((w[6] >> 11) | (w[7] << 21)) & BP_M29;
    r.v[8] = w[7] >> 8;
    return r;
}
// canonicalises first: the words are the unique value in [0, p)
BP_HD fp_words fp_to_words(const fp& a) {
    fp c = fp_canon(a);
    const uint32_t* v = c.v;
    fp_words s;
    s.w[0] = v[0] | (v[1] << 29);
    s.w[1] = (v[1] >> 3) | (v[2] << 26);
    s.w[2] = (v[2] >> 6) | (v[3] << 23);
    s.w[3] = (v[3] >> 9) | (v[4] << 20);
    s.w[4] = (v[4] >> 12) | (v[5] << 17);
    s.w[5] = (v[5] >> 15) | (v[6] << 14);
    s.w[6] = (v[6] >> 18) | (v[7] << 11);
    s.w[7] = (v[7] >> 21) | (v[8] << 8);
    return s;
}
// words < p ?  (for validating untrusted input)
BP_HD bool fp_words_is_canonical(const fp_words& s) {
    // s + (2^32 + 977) must not carry out of 256 bits
    uint64_t c = (uint64_t)s.w[0] + BP_P_C0;
    c >>= 32;
    c += (uint64_t)s.w[1] + 1u;
    c >>= 32;
#pragma unroll
    for (int i = 2; i < 8; i++) {
        c += s.w[i];
        c >>= 32;
    }
    return c == 0;
}
BP_HD bool fp_words_is_zero(const fp_words& s) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= s.w[i];
    return o == 0;
}

// 32 big-endian bytes <-> words / fp.  from_be does not range-check (fp_words_is_canonical does).
BP_HD fp_words fp_words_from_be(const uint8_t* b) {
    fp_words r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint8_t* q = b + 4 * (7 - i);
        r.w[i] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | (uint32_t)q[3];
    }
    return r;
}
BP_HD void fp_words_to_be(uint8_t* b, const fp_words& a) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint8_t* q = b + 4 * (7 - i);
        q[0] = (uint8_t)(a.w[i] >> 24);
        q[1] = (uint8_t)(a.w[i] >> 16);
        q[2] = (uint8_t)(a.w[i] >> 8);
        q[3] = (uint8_t)(a.w[i]);
    }
}
BP_HD fp fp_from_be(const uint8_t* b) { return fp_from_words(fp_words_from_be(b)); }
BP_HD void fp_to_be(uint8_t* b, const fp& a) { fp_words_to_be(b, fp_to_words(a)); }
BP_HD bool fp_be_is_canonical(const uint8_t* b) { return fp_words_is_canonical(fp_words_from_be(b)); }

// ---- fp_inv_var: inversion by Bernstein-Yang division steps ("safegcd"), variable time ----------------
// For the single-lane tails (the last step of every MSM, one per call): a lone wave pays ~0.35 us per dependent
// field multiplication, so Fermat's 270 of them are ~90 us; 18-25 rounds of 30 division steps on 30-bit signed limbs
// are a third of that.  NOT for the throughput kernels: iteration counts and branches differ per lane, and a
// wavefront pays for the union (those keep fp_inv).
// Invariants (Bernstein, Yang: "Fast constant-time gcd computation and modular inversion", 2019; the limb layout and
// the d/e update follow the published 32-bit formulation used by libsecp256k1's modinv32): f odd, f = d*x, g = e*x
// (mod p), starting from f = p, g = x, d = 0, e = 1; divstep(delta, f, g) = (1 - delta, g, (g - f)/2) if delta > 0 and
// g odd, else (1 + delta, f, (g + (g mod 2) f)/2).  When g reaches 0, f = +-1 and x^-1 = +-d.
struct fp_s30 {
    int32_t v[9];  // value = sum v[i] 2^(30 i); v[0..7] in [0, 2^30) once propagated, v[8] carries the sign
};

// 30 division steps on the low words; t = (u, v, q, r) with 2^30 f' = u f + v g, 2^30 g' = q f + r g.
BP_HD int fp_divsteps30(int delta, uint32_t f, uint32_t g, int32_t t[4]) {
    int32_t u = 1, v = 0, q = 0, r = 1;
    for (int i = 0; i < 30; i++) {
        if (g & 1u) {
            if (delta > 0) {
                delta = 1 - delta;
                const uint32_t nf = g;
                g = (g - f) >> 1;
                f = nf;
                const int32_t nu = q * 2, nv = r * 2;
                q = q - u;
                r = r - v;
                u = nu;
                v = nv;
            } else {
                delta = 1 + delta;
                g = (g + f) >> 1;
                q = q + u;
                r = r + v;
                u *= 2;
                v *= 2;
            }
        } else {
            delta = 1 + delta;
            g >>= 1;
            u *= 2;
            v *= 2;
        }
    }
    t[0] = u;
    t[1] = v;
    t[2] = q;
    t[3] = r;
    return delta;
}

BP_HD void fp_s30_update_fg(fp_s30& f, fp_s30& g, const int32_t t[4]) {
    const int64_t u = t[0], v = t[1], q = t[2], r = t[3];
    int64_t cf = u * f.v[0] + v * g.v[0];
    int64_t cg = q * f.v[0] + r * g.v[0];
    cf >>= 30;  // the low 30 bits are zero by construction
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        const int64_t fi = f.v[i], gi = g.v[i];
        cf += u * fi + v * gi;
        cg += q * fi + r * gi;
        f.v[i - 1] = (int32_t)(cf & 0x3fffffff);
        g.v[i - 1] = (int32_t)(cg & 0x3fffffff);
        cf >>= 30;
        cg >>= 30;
    }
    f.v[8] = (int32_t)cf;
    g.v[8] = (int32_t)cg;
}

// The modulus as 30-bit limbs and its inverse mod 2^30: p here, the group order q in fq.h (s30_mod_q).
struct s30_mod_p {
    static BP_HD int32_t limb(int i) {  // p = 2^256 - 2^32 - 977
        return i == 0 ? 0x3ffffc2f : (i == 1 ? 0x3ffffffb : (i < 8 ? 0x3fffffff : 0xffff));
    }
    static BP_HD uint32_t inv30() { return 0x2ddacacfu; }  // p^-1 mod 2^30
};

// (d, e) <- (u d + v e, q d + r e) / 2^30 mod M, kept in (-2M, M)
template <class M>
BP_HD void s30_update_de(fp_s30& d, fp_s30& e, const int32_t t[4]) {
    const uint32_t M_INV30 = M::inv30();
    const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
    const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
    int32_t md = (u & sd) + (v & se);
    int32_t me = (q & sd) + (r & se);
    int64_t cd = (int64_t)u * d.v[0] + (int64_t)v * e.v[0];
    int64_t ce = (int64_t)q * d.v[0] + (int64_t)r * e.v[0];
    md -= (int32_t)((M_INV30 * (uint32_t)cd + (uint32_t)md) & 0x3fffffffu);
    me -= (int32_t)((M_INV30 * (uint32_t)ce + (uint32_t)me) & 0x3fffffffu);
    cd += (int64_t)M::limb(0) * md;
    ce += (int64_t)M::limb(0) * me;
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        const int64_t di = d.v[i], ei = e.v[i];
        cd += (int64_t)u * di + (int64_t)v * ei + (int64_t)M::limb(i) * md;
        ce += (int64_t)q * di + (int64_t)r * ei + (int64_t)M::limb(i) * me;
        d.v[i - 1] = (int32_t)(cd & 0x3fffffff);
        e.v[i - 1] = (int32_t)(ce & 0x3fffffff);
        cd >>= 30;
        ce >>= 30;
    }
    d.v[8] = (int32_t)cd;
    e.v[8] = (int32_t)ce;
}

// x^-1 mod M for a canonical x in [0, M) given and returned as 8 little-endian 32-bit words; 0 -> 0.
template <class M>
BP_HD void s30_inv_words(const uint32_t aw[8], uint32_t rw[8]) {
    fp_s30 f, g, d, e;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        f.v[i] = M::limb(i);
        d.v[i] = 0;
        e.v[i] = 0;
        // bits [30 i, 30 i + 30) of the 256-bit value
        const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
        uint64_t two = aw[w];
        if (w + 1 < 8) two |= (uint64_t)aw[w + 1] << 32;
        g.v[i] = (int32_t)((uint32_t)(two >> sh) & 0x3fffffffu);
    }
    e.v[0] = 1;
    int delta = 1;
    // at most 741 division steps are needed for 256-bit inputs (25 rounds); the bound below is only a guard
    for (int it = 0; it < 40; it++) {
        int32_t gz = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) gz |= g.v[i];
        if (gz == 0) break;
        int32_t t[4];
        delta = fp_divsteps30(delta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
        s30_update_de<M>(d, e, t);
        fp_s30_update_fg(f, g, t);
    }
    // f = +-1: x^-1 = f * d.  Normalise d from (-2M, M) to [0, M).
    const bool neg = f.v[8] < 0;
    int64_t c = 0;
    int32_t o[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        c += neg ? -(int64_t)d.v[i] : (int64_t)d.v[i];
        if (i < 8) {
            o[i] = (int32_t)(c & 0x3fffffff);
            c >>= 30;
        } else {
            o[8] = (int32_t)c;  // the top limb keeps the sign
        }
    }
    for (int pass = 0; pass < 3; pass++) {  // value in (-2M, 2M): add or subtract M until it is in [0, M)
        const bool isneg = o[8] < 0;
        bool ge = !isneg;  // compare with M from the top limb down
        if (ge) {
            for (int i = 8; i >= 0; i--) {
                if (o[i] != M::limb(i)) {
                    ge = o[i] > M::limb(i);
                    break;
                }
            }
        }
        if (!isneg && !ge) break;
        int64_t cc = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            cc += (int64_t)o[i] + (isneg ? (int64_t)M::limb(i) : -(int64_t)M::limb(i));
            if (i < 8) {
                o[i] = (int32_t)(cc & 0x3fffffff);
                cc >>= 30;
            } else {
                o[8] = (int32_t)cc;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        // word k = bits [32 k, 32 k + 32) from the 30-bit limbs
        const int bit = 32 * k, l = bit / 30, sh = bit % 30;
        uint64_t acc = (uint64_t)(uint32_t)o[l] >> sh;
        acc |= (uint64_t)(uint32_t)o[l + 1] << (30 - sh);  // sh <= 14: two limbs cover the word
        rw[k] = (uint32_t)acc;
    }
}

BP_HD fp fp_inv_var(const fp& a) {
    const fp_words aw = fp_to_words(a);  // canonical value in [0, p)
    fp_words rw;
    s30_inv_words<s30_mod_p>(aw.w, rw.w);
    return fp_from_words(rw);
}

}  // namespace bp
