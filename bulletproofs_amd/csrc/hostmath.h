// Host-side helpers for the protocol layer (proto.cpp): scalar class over fq.h, curv-compatible
// transcript (SHA-256, DigestExt byte formats of SURVEY.md appendix A), SHA-512 and
// `generate_random_point` (/root/reference/src/proofs/range_proof.rs:785-808).  CPU code: this is
// the part of the reference that stays on the host in a Rust deployment; every group operation is
// delegated to the HIP kernels.
#pragma once
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "ec.h"
#include "fq.h"
#include "sha256.h"

namespace bp {

// ---- scalar mod q on the host ---------------------------------------------------------------------
// 4 x 64-bit limbs kept in the Montgomery domain (R = 2^256), always canonical (< q), so +, -, ==, zero
// tests work on the representation and a product is ONE CIOS pass over unsigned __int128 (~30 ns; the
// plain-domain 8 x 32 fq_mul of fq.h costs two passes and ~300 ns).  The prover's per-round coefficient
// updates (ipa_prove_fixed in proto.cpp: 4nm products per round) made this the host's hot loop.
// Bytes at the boundary are converted on the way in / out and are the reference's canonical values.
struct Fq {
    uint64_t m[4];
    typedef unsigned __int128 u128;
    static constexpr uint64_t Q0 = 0xBFD25E8CD0364141ULL, Q1 = 0xBAAEDCE6AF48A03BULL, Q2 = 0xFFFFFFFFFFFFFFFEULL, Q3 = 0xFFFFFFFFFFFFFFFFULL;
    static constexpr uint64_t QINV = 0x4B0DFF665588B13FULL;  // -q^-1 mod 2^64
    static void mont_mul(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
        const uint64_t q[4] = {Q0, Q1, Q2, Q3};
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 4; j++) {
                c += (u128)a[i] * b[j] + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[4] = (uint64_t)c;
            t[5] = (uint64_t)(c >> 64);
            const uint64_t mm = t[0] * QINV;
            c = (u128)mm * q[0] + t[0];
            c >>= 64;
            for (int j = 1; j < 4; j++) {
                c += (u128)mm * q[j] + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[3] = (uint64_t)c;
            t[4] = t[5] + (uint64_t)(c >> 64);
        }
        reduce_once(r, t, t[4]);
    }
    // r = (carry * 2^256 + t) mod q for a value < 2q
    static void reduce_once(uint64_t r[4], const uint64_t t[4], uint64_t carry) {
        const uint64_t q[4] = {Q0, Q1, Q2, Q3};
        uint64_t s[4];
        u128 bw = 0;
        for (int i = 0; i < 4; i++) {
            u128 d = (u128)t[i] - q[i] - (uint64_t)bw;
            s[i] = (uint64_t)d;
            bw = (d >> 64) & 1;
        }
        const bool take = carry || !bw;
        for (int i = 0; i < 4; i++) r[i] = take ? s[i] : t[i];
    }
    Fq() { m[0] = m[1] = m[2] = m[3] = 0; }
    explicit Fq(uint64_t x) {
        const uint64_t R2[4] = {0x896CF21467D7D140ULL, 0x741496C20E7CF878ULL, 0xE697F5E45BCD07C6ULL, 0x9D671CD581C69BC5ULL};
        const uint64_t a[4] = {x, 0, 0, 0};
        mont_mul(m, a, R2);
    }
    static Fq from_be(const uint8_t* b) {  // any 256-bit value, reduced like Scalar::from(&BigInt)
        const uint64_t R2[4] = {0x896CF21467D7D140ULL, 0x741496C20E7CF878ULL, 0xE697F5E45BCD07C6ULL, 0x9D671CD581C69BC5ULL};
        fq k = fq_from_be_reduce(b);
        uint64_t a[4];
        for (int i = 0; i < 4; i++) a[i] = (uint64_t)k.v[2 * i] | ((uint64_t)k.v[2 * i + 1] << 32);
        Fq r;
        mont_mul(r.m, a, R2);
        return r;
    }
    void to_be(uint8_t* b) const {
        const uint64_t one[4] = {1, 0, 0, 0};
        uint64_t a[4];
        mont_mul(a, m, one);
        fq k;
        for (int i = 0; i < 4; i++) {
            k.v[2 * i] = (uint32_t)a[i];
            k.v[2 * i + 1] = (uint32_t)(a[i] >> 32);
        }
        fq_to_be(b, k);
    }
    bool is_zero() const { return (m[0] | m[1] | m[2] | m[3]) == 0; }
    bool operator==(const Fq& o) const { return ((m[0] ^ o.m[0]) | (m[1] ^ o.m[1]) | (m[2] ^ o.m[2]) | (m[3] ^ o.m[3])) == 0; }
    Fq operator+(const Fq& o) const {
        uint64_t t[4];
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)m[i] + o.m[i];
            t[i] = (uint64_t)c;
            c >>= 64;
        }
        Fq r;
        reduce_once(r.m, t, (uint64_t)c);
        return r;
    }
    Fq operator-(const Fq& o) const {
        const uint64_t q[4] = {Q0, Q1, Q2, Q3};
        Fq r;
        u128 bw = 0;
        for (int i = 0; i < 4; i++) {
            u128 d = (u128)m[i] - o.m[i] - (uint64_t)bw;
            r.m[i] = (uint64_t)d;
            bw = (d >> 64) & 1;
        }
        if (bw) {
            u128 c = 0;
            for (int i = 0; i < 4; i++) {
                c += (u128)r.m[i] + q[i];
                r.m[i] = (uint64_t)c;
                c >>= 64;
            }
        }
        return r;
    }
    Fq operator-() const { return Fq() - *this; }
    Fq operator*(const Fq& o) const {
        Fq r;
        mont_mul(r.m, m, o.m);
        return r;
    }
    Fq sqr_n(int n) const {
        Fq r = *this;
        for (int i = 0; i < n; i++) r = r * r;
        return r;
    }
    // Inverse by Bernstein-Yang division steps (fp.h s30_inv_words with the modulus q, the code the device's fqm_inv_var runs):
    // ~2 us on a host core against ~12 us for the exponentiation below, and a single-proof prover inverts one challenge per
    // round between two GPU calls.  0 -> 0 (callers check for zero first: the reference unwraps).
    Fq inv() const {
        const uint64_t one[4] = {1, 0, 0, 0};
        const uint64_t R2[4] = {0x896CF21467D7D140ULL, 0x741496C20E7CF878ULL, 0xE697F5E45BCD07C6ULL, 0x9D671CD581C69BC5ULL};
        uint64_t plain[4];
        mont_mul(plain, m, one);
        uint32_t aw[8], rw[8];
        for (int i = 0; i < 4; i++) {
            aw[2 * i] = (uint32_t)plain[i];
            aw[2 * i + 1] = (uint32_t)(plain[i] >> 32);
        }
        s30_inv_words<s30_mod_q>(aw, rw);
        uint64_t r64[4];
        for (int i = 0; i < 4; i++) r64[i] = (uint64_t)rw[2 * i] | ((uint64_t)rw[2 * i + 1] << 32);
        Fq r;
        mont_mul(r.m, r64, R2);
        return r;
    }
    // a^(q-2): the addition chain of fq_mont_inv (fq.h): 2^127 - 1 ones, then a sliding window over the low
    // 129 bits.  Caller guarantees non-zero (the reference unwraps).
    Fq inv_fermat() const {
        static const uint8_t nsqr[26] = {5, 3, 4, 4, 5, 2, 5, 6, 5, 4, 3, 6, 10, 4, 5, 4, 5, 6, 4, 5, 6, 10, 4, 9, 4, 1};
        static const uint8_t odd[26] = {11, 5, 5, 7, 13, 3, 7, 13, 11, 13, 1, 5, 7, 7, 15, 15, 9, 11, 13, 3, 13, 13, 9, 9, 15, 1};
        Fq tbl[8];
        tbl[0] = *this;
        Fq a2 = *this * *this;
        for (int i = 1; i < 8; i++) tbl[i] = tbl[i - 1] * a2;
        Fq x3 = tbl[3];
        Fq x6 = x3.sqr_n(3) * x3;
        Fq x12 = x6.sqr_n(6) * x6;
        Fq x24 = x12.sqr_n(12) * x12;
        Fq x48 = x24.sqr_n(24) * x24;
        Fq x96 = x48.sqr_n(48) * x48;
        Fq x120 = x96.sqr_n(24) * x24;
        Fq x126 = x120.sqr_n(6) * x6;
        Fq acc = x126 * x126 * *this;
        for (int s = 0; s < 26; s++) acc = acc.sqr_n(nsqr[s]) * tbl[odd[s] >> 1];
        return acc;
    }
    Fq pow(uint64_t e) const {
        Fq r(1), b = *this;
        while (e) {
            if (e & 1) r = r * b;
            b = b * b;
            e >>= 1;
        }
        return r;
    }
};

// ---- SHA-512 (generator labels only) -------------------------------------------------------------
inline void sha512(const uint8_t* data, size_t len, uint8_t out[64]) {
    static const uint64_t K[80] = {
        0x428a2f98d728ae22ULL, 0x7137449123ef65cdULL, 0xb5c0fbcfec4d3b2fULL, 0xe9b5dba58189dbbcULL, 0x3956c25bf348b538ULL, 0x59f111f1b605d019ULL,
        0x923f82a4af194f9bULL, 0xab1c5ed5da6d8118ULL, 0xd807aa98a3030242ULL, 0x12835b0145706fbeULL, 0x243185be4ee4b28cULL, 0x550c7dc3d5ffb4e2ULL,
        0x72be5d74f27b896fULL, 0x80deb1fe3b1696b1ULL, 0x9bdc06a725c71235ULL, 0xc19bf174cf692694ULL, 0xe49b69c19ef14ad2ULL, 0xefbe4786384f25e3ULL,
        0x0fc19dc68b8cd5b5ULL, 0x240ca1cc77ac9c65ULL, 0x2de92c6f592b0275ULL, 0x4a7484aa6ea6e483ULL, 0x5cb0a9dcbd41fbd4ULL, 0x76f988da831153b5ULL,
        0x983e5152ee66dfabULL, 0xa831c66d2db43210ULL, 0xb00327c898fb213fULL, 0xbf597fc7beef0ee4ULL, 0xc6e00bf33da88fc2ULL, 0xd5a79147930aa725ULL,
        0x06ca6351e003826fULL, 0x142929670a0e6e70ULL, 0x27b70a8546d22ffcULL, 0x2e1b21385c26c926ULL, 0x4d2c6dfc5ac42aedULL, 0x53380d139d95b3dfULL,
        0x650a73548baf63deULL, 0x766a0abb3c77b2a8ULL, 0x81c2c92e47edaee6ULL, 0x92722c851482353bULL, 0xa2bfe8a14cf10364ULL, 0xa81a664bbc423001ULL,
        0xc24b8b70d0f89791ULL, 0xc76c51a30654be30ULL, 0xd192e819d6ef5218ULL, 0xd69906245565a910ULL, 0xf40e35855771202aULL, 0x106aa07032bbd1b8ULL,
        0x19a4c116b8d2d0c8ULL, 0x1e376c085141ab53ULL, 0x2748774cdf8eeb99ULL, 0x34b0bcb5e19b48a8ULL, 0x391c0cb3c5c95a63ULL, 0x4ed8aa4ae3418acbULL,
        0x5b9cca4f7763e373ULL, 0x682e6ff3d6b2b8a3ULL, 0x748f82ee5defb2fcULL, 0x78a5636f43172f60ULL, 0x84c87814a1f0ab72ULL, 0x8cc702081a6439ecULL,
        0x90befffa23631e28ULL, 0xa4506cebde82bde9ULL, 0xbef9a3f7b2c67915ULL, 0xc67178f2e372532bULL, 0xca273eceea26619cULL, 0xd186b8c721c0c207ULL,
        0xeada7dd6cde0eb1eULL, 0xf57d4f7fee6ed178ULL, 0x06f067aa72176fbaULL, 0x0a637dc5a2c898a6ULL, 0x113f9804bef90daeULL, 0x1b710b35131c471bULL,
        0x28db77f523047d84ULL, 0x32caab7b40c72493ULL, 0x3c9ebe0a15c9bebcULL, 0x431d67c49c100d4cULL, 0x4cc5d4becb3e42b6ULL, 0x597f299cfc657e2aULL,
        0x5fcb6fab3ad6faecULL, 0x6c44198c4a475817ULL};
    uint64_t h[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                     0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    std::vector<uint8_t> msg(data, data + len);
    msg.push_back(0x80);
    while (msg.size() % 128 != 112) msg.push_back(0);
    for (int i = 0; i < 8; i++) msg.push_back(0);
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 7; i >= 0; i--) msg.push_back((uint8_t)(bits >> (8 * i)));
    auto rotr = [](uint64_t x, int n) { return (x >> n) | (x << (64 - n)); };
    for (size_t off = 0; off < msg.size(); off += 128) {
        uint64_t w[80];
        for (int i = 0; i < 16; i++) {
            uint64_t v = 0;
            for (int j = 0; j < 8; j++) v = (v << 8) | msg[off + 8 * i + j];
            w[i] = v;
        }
        for (int i = 16; i < 80; i++) {
            uint64_t s0 = rotr(w[i - 15], 1) ^ rotr(w[i - 15], 8) ^ (w[i - 15] >> 7);
            uint64_t s1 = rotr(w[i - 2], 19) ^ rotr(w[i - 2], 61) ^ (w[i - 2] >> 6);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint64_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 80; i++) {
            uint64_t S1 = rotr(e, 14) ^ rotr(e, 18) ^ rotr(e, 41);
            uint64_t ch = (e & f) ^ (~e & g);
            uint64_t t1 = hh + S1 + ch + K[i] + w[i];
            uint64_t S0 = rotr(a, 28) ^ rotr(a, 34) ^ rotr(a, 39);
            uint64_t mj = (a & b) ^ (a & c) ^ (b & c);
            uint64_t t2 = S0 + mj;
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
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(h[i] >> (56 - 8 * j));
}

// ---- curv DigestExt over SHA-256 -----------------------------------------------------------------
struct Transcript {
    sha256_ctx c;
    Transcript() { sha256_init(c); }
    // chain_points: 65-byte uncompressed encoding of a 64-byte x||y point
    Transcript& point(const uint8_t* xy64) {
        uint8_t four = 4;
        sha256_update(c, &four, 1);
        sha256_update(c, xy64, 64);
        return *this;
    }
    // chain_bigint of a value given as 32 big-endian bytes: minimal magnitude, zero -> 0x00
    Transcript& bigint32(const uint8_t* be32) {
        int i = 0;
        while (i < 31 && be32[i] == 0) i++;
        sha256_update(c, be32 + i, 32 - i);
        return *this;
    }
    Transcript& scalar(const Fq& s) {
        uint8_t b[32];
        s.to_be(b);
        return bigint32(b);
    }
    // result_scalar: try-and-increment with a 4-byte big-endian counter
    Fq result_scalar() const {
        for (uint32_t i = 0;; i++) {
            sha256_ctx t = c;
            uint8_t ctr[4] = {(uint8_t)(i >> 24), (uint8_t)(i >> 16), (uint8_t)(i >> 8), (uint8_t)i};
            sha256_update(t, ctr, 4);
            uint8_t d[32];
            sha256_final(t, d);
            if (fq_be_is_canonical(d)) return Fq::from_be(d);
        }
    }
};

// ---- base field on 4x64-bit limbs (host only): decompression of proof points ------------------------
// The single-proof verifiers parse 4 + 2k compressed points per proof on the host (curv `Point::from_bytes`); with
// the portable 9x29 code of fp.h (written for the GPU's 32-bit multiplier) one square root costs ~25 us of a host
// core, a third of a verify call.  p = 2^256 - C, C = 2^32 + 977: schoolbook product, two folds, one subtraction.
struct Fp64 {
    uint64_t v[4];
    static void mul(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
        typedef unsigned __int128 u128;
        uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 4; j++) {
                c += (u128)a[i] * b[j] + t[i + j];
                t[i + j] = (uint64_t)c;
                c >>= 64;
            }
            t[i + 4] = (uint64_t)c;
        }
        const uint64_t C = 0x1000003D1ULL;
        // fold 1: lo + hi * C  (5 limbs)
        uint64_t f[5];
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)t[4 + i] * C + t[i];
            f[i] = (uint64_t)c;
            c >>= 64;
        }
        f[4] = (uint64_t)c;
        // fold 2: f[4] * C into the low limbs (f[4] < 2^34)
        c = (u128)f[4] * C;
        for (int i = 0; i < 4; i++) {
            c += f[i];
            f[i] = (uint64_t)c;
            c >>= 64;
        }
        if ((uint64_t)c) {  // wrapped past 2^256 once more: add C (cannot carry again: the low limbs are tiny then)
            u128 d = (u128)f[0] + C;
            f[0] = (uint64_t)d;
            d >>= 64;
            for (int i = 1; i < 4 && (uint64_t)d; i++) {
                d += f[i];
                f[i] = (uint64_t)d;
                d >>= 64;
            }
        }
        // final: subtract p if >= p
        const uint64_t P[4] = {0xFFFFFFFEFFFFFC2FULL, ~0ULL, ~0ULL, ~0ULL};
        bool ge = true;
        for (int i = 3; i >= 0; i--) {
            if (f[i] != P[i]) {
                ge = f[i] > P[i];
                break;
            }
        }
        if (ge) {
            u128 bw = 0;
            for (int i = 0; i < 4; i++) {
                u128 d = (u128)f[i] - P[i] - (uint64_t)bw;
                f[i] = (uint64_t)d;
                bw = (d >> 64) & 1;
            }
        }
        for (int i = 0; i < 4; i++) r[i] = f[i];
    }
    Fp64 operator*(const Fp64& o) const {
        Fp64 r;
        mul(r.v, v, o.v);
        return r;
    }
    Fp64 sqr_n(int n) const {
        Fp64 r = *this;
        for (int i = 0; i < n; i++) r = r * r;
        return r;
    }
    bool operator==(const Fp64& o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2] && v[3] == o.v[3]; }
    static Fp64 from_be(const uint8_t* b) {
        Fp64 r;
        for (int i = 0; i < 4; i++) {
            uint64_t w = 0;
            for (int j = 0; j < 8; j++) w = (w << 8) | b[8 * (3 - i) + j];
            r.v[i] = w;
        }
        return r;
    }
    void to_be(uint8_t* b) const {
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 8; j++) b[8 * (3 - i) + j] = (uint8_t)(v[i] >> (56 - 8 * j));
    }
    bool lt_p() const {
        const uint64_t P[4] = {0xFFFFFFFEFFFFFC2FULL, ~0ULL, ~0ULL, ~0ULL};
        for (int i = 3; i >= 0; i--)
            if (v[i] != P[i]) return v[i] < P[i];
        return false;
    }
    // a^((p+1)/4): the addition chain of fp_sqrt_candidate (fp.h)
    Fp64 sqrt_candidate() const {
        const Fp64& a = *this;
        Fp64 x2 = a.sqr_n(1) * a;
        Fp64 x3 = x2.sqr_n(1) * a;
        Fp64 x6 = x3.sqr_n(3) * x3;
        Fp64 x9 = x6.sqr_n(3) * x3;
        Fp64 x11 = x9.sqr_n(2) * x2;
        Fp64 x22 = x11.sqr_n(11) * x11;
        Fp64 x44 = x22.sqr_n(22) * x22;
        Fp64 x88 = x44.sqr_n(44) * x44;
        Fp64 x176 = x88.sqr_n(88) * x88;
        Fp64 x220 = x176.sqr_n(44) * x44;
        Fp64 x223 = x220.sqr_n(3) * x3;
        Fp64 t = x223.sqr_n(23) * x22;
        t = t.sqr_n(6) * x2;
        return t.sqr_n(2);
    }
};

// ---- host point helpers (validation / compression only; no group law on the host) -----------------
// 33-byte SEC1 compressed -> x || y (curv `Point::from_bytes`); same accept/reject rule as affine_decompress (ec.h)
inline bool host_decompress(const uint8_t* b33, uint8_t out_xy[64]) {
    if (b33[0] != 2 && b33[0] != 3) return false;
    Fp64 x = Fp64::from_be(b33 + 1);
    if (!x.lt_p()) return false;
    Fp64 rhs = x * x * x;
    {  // + 7
        typedef unsigned __int128 u128;
        u128 c = (u128)rhs.v[0] + 7;
        rhs.v[0] = (uint64_t)c;
        c >>= 64;
        for (int i = 1; i < 4; i++) {
            c += rhs.v[i];
            rhs.v[i] = (uint64_t)c;
            c >>= 64;
        }
        // a carry out of 2^256 or a value in [p, 2^256) can only follow rhs >= p - 7: renormalise through a product by one
        if ((uint64_t)c || !rhs.lt_p()) {
            // rhs (+ 2^256 if carry) mod p: add C for the carry, then reduce
            if ((uint64_t)c) {
                u128 d = (u128)rhs.v[0] + 0x1000003D1ULL;
                rhs.v[0] = (uint64_t)d;
                d >>= 64;
                for (int i = 1; i < 4; i++) {
                    d += rhs.v[i];
                    rhs.v[i] = (uint64_t)d;
                    d >>= 64;
                }
            }
            Fp64 one;
            one.v[0] = 1;
            one.v[1] = one.v[2] = one.v[3] = 0;
            rhs = rhs * one;
        }
    }
    Fp64 y = rhs.sqrt_candidate();
    if (!(y * y == rhs)) return false;
    if ((y.v[0] & 1) != (uint64_t)(b33[0] & 1)) {  // y = p - y  (y != 0: the group has prime order)
        typedef unsigned __int128 u128;
        const uint64_t P[4] = {0xFFFFFFFEFFFFFC2FULL, ~0ULL, ~0ULL, ~0ULL};
        u128 bw = 0;
        for (int i = 0; i < 4; i++) {
            u128 d = (u128)P[i] - y.v[i] - (uint64_t)bw;
            y.v[i] = (uint64_t)d;
            bw = (d >> 64) & 1;
        }
    }
    x.to_be(out_xy);
    y.to_be(out_xy + 32);
    return true;
}
// the portable-field version (fp.h code path shared with the GPU), kept for the host self-test
inline bool host_decompress_portable(const uint8_t* b33, uint8_t out_xy[64]) {
    affine a;
    if (!affine_decompress(a, b33)) return false;
    affine_to_be64(out_xy, a);
    return true;
}
inline void host_compress(const uint8_t* xy64, uint8_t out33[33]) {
    out33[0] = (uint8_t)(2 + (xy64[63] & 1));
    memcpy(out33 + 1, xy64, 32);
}
inline bool host_is_inf(const uint8_t* xy64) {
    for (int i = 0; i < 64; i++)
        if (xy64[i]) return false;
    return true;
}

// minimal big-endian bytes of a non-negative integer given as big-endian bytes (BigInt::to_bytes)
inline std::vector<uint8_t> strip_leading_zeros(const uint8_t* b, size_t n) {
    size_t i = 0;
    while (i + 1 < n && b[i] == 0) i++;
    return std::vector<uint8_t>(b + i, b + n);
}
// (int(bytes) * 2) mod q as minimal big-endian bytes
inline std::vector<uint8_t> times_two_mod_q(const std::vector<uint8_t>& bytes) {
    Fq acc, c256(256);
    for (uint8_t byte : bytes) acc = acc * c256 + Fq(byte);
    acc = acc + acc;
    uint8_t be[32];
    acc.to_be(be);
    return strip_leading_zeros(be, 32);
}
// range_proof.rs:785-808
inline void generate_random_point(std::vector<uint8_t> bytes, uint8_t out_xy[64]) {
    for (;;) {
        uint8_t buf[33];
        memset(buf, 0, 33);
        buf[0] = 2;
        size_t take = bytes.size() > 32 ? 32 : bytes.size();
        memcpy(buf + 1, bytes.data(), take);
        if (host_decompress(buf, out_xy)) return;
        bytes = times_two_mod_q(bytes);
    }
}
// generate_random_point(to_bytes(Sha512(chain_bigint(label)).result_bigint())) for a small label
inline void label_point(const std::vector<uint8_t>& label_min_be, uint8_t out_xy[64]) {
    uint8_t d[64];
    sha512(label_min_be.data(), label_min_be.size(), d);
    generate_random_point(strip_leading_zeros(d, 64), out_xy);
}
inline std::vector<uint8_t> u64_min_be(uint64_t x) {
    uint8_t b[8];
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(x >> (56 - 8 * i));
    return strip_leading_zeros(b, 8);
}

}  // namespace bp
