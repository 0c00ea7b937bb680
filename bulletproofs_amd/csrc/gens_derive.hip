// Generator derivation on the GPU: the synthetic-input recipe of the reference's tests and
// benches, one lane per generator.
//
//   g_vec[i] = grp(SHA512(be(seed + i))), h_vec[i] = grp(SHA512(be(seed + n + i)))     /root/reference/src/proofs/range_proof.rs:821-843,
//   G = secp256k1 generator, H = grp(SHA512(be(1)))                                     benches/range_proof.rs:43-67
//   StatementRP::generate_bases: the same with H = grp(SHA512(be(seed - 1)))            range_proof_wip.rs:48-91
//   grp = generate_random_point (range_proof.rs:785-808): try to decompress 0x02 || bytes[0..32] (left-aligned,
//   zero-padded when shorter); on failure bytes <- to_bytes(int(bytes) * 2 mod q) and try again.
//
// be() / to_bytes are curv's minimal big-endian encodings: SHA-512 hashes the label
// without leading zero bytes, and a digest that starts with zero bytes loses them before the 32-byte window is
// taken.  The host recipe (hostmath.h: label_point) stays as the reference implementation of the same steps; the
// GPU test compares the two byte for byte.
#include <hip/hip_runtime.h>

#include "ec.h"
#include "fq.h"
#include "kernels.h"

namespace bp {

__constant__ uint64_t SHA512_K[80] = {
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

__device__ __forceinline__ uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

// SHA-512 of the minimal big-endian bytes of a 64-bit label (1..8 bytes: always one block); digest as 8 words.
__device__ void sha512_label(uint64_t label, uint64_t h[8]) {
    int len = 1;
    while (len < 8 && (label >> (8 * len)) != 0) len++;
    uint64_t w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = 0;
    // message bytes: label (len bytes, big-endian), 0x80, zeros, 128-bit length
    w[0] = (label << (64 - 8 * len));
    if (len < 8)
        w[0] |= (uint64_t)0x80 << (56 - 8 * len);
    else
        w[1] = (uint64_t)0x80 << 56;
    w[15] = (uint64_t)len * 8;
    uint64_t s[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                     0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    uint64_t a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5], g = s[6], hh = s[7];
    for (int i = 0; i < 80; i++) {
        uint64_t wi;
        if (i < 16) {
            wi = w[i];
        } else {
            const uint64_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            const uint64_t s0 = rotr64(w15, 1) ^ rotr64(w15, 8) ^ (w15 >> 7);
            const uint64_t s1 = rotr64(w2, 19) ^ rotr64(w2, 61) ^ (w2 >> 6);
            wi = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
            w[i & 15] = wi;
        }
        const uint64_t S1 = rotr64(e, 14) ^ rotr64(e, 18) ^ rotr64(e, 41);
        const uint64_t ch = (e & f) ^ (~e & g);
        const uint64_t t1 = hh + S1 + ch + SHA512_K[i] + wi;
        const uint64_t S0 = rotr64(a, 28) ^ rotr64(a, 34) ^ rotr64(a, 39);
        const uint64_t mj = (a & b) ^ (a & c) ^ (b & c);
        const uint64_t t2 = S0 + mj;
        hh = g;
        g = f;
        f = e;
        e = d + t1;
        d = c;
        c = b;
        b = a;
        a = t1 + t2;
    }
    h[0] = s[0] + a;
    h[1] = s[1] + b;
    h[2] = s[2] + c;
    h[3] = s[3] + d;
    h[4] = s[4] + e;
    h[5] = s[5] + f;
    h[6] = s[6] + g;
    h[7] = s[7] + hh;
}

// generate_random_point over `bytes[0 .. len)` (len <= 64, no leading zero byte unless len == 1)
__device__ affine grp_from_bytes(uint8_t* bytes, int len) {
    for (;;) {
        uint8_t buf[33];
        buf[0] = 2;
        for (int i = 0; i < 32; i++) buf[1 + i] = i < len ? bytes[i] : 0;
        affine p;
        if (affine_decompress(p, buf)) return p;
        // bytes <- to_bytes(int(bytes) * 2 mod q)
        fq acc = fq_zero();
        const fq c256 = fq_from_u32(256);
        for (int i = 0; i < len; i++) acc = fq_add(fq_mul(acc, c256), fq_from_u32(bytes[i]));
        acc = fq_add(acc, acc);
        uint8_t be[32];
        fq_to_be(be, acc);
        int z = 0;
        while (z < 31 && be[z] == 0) z++;
        len = 32 - z;
        for (int i = 0; i < len; i++) bytes[i] = be[z + i];
    }
}

// out[0 .. nm) = g_vec, out[nm .. 2nm) = h_vec, out[2nm] = G, out[2nm + 1] = H  (the order of bp_gens)
__global__ __launch_bounds__(64) void k_gens_derive(int kind, uint64_t seed, uint64_t n_bits, size_t nm, affine_mem* __restrict__ out) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= 2 * nm + 2) return;
    if (p == 2 * nm) {
        const uint32_t GX[8] = {0x16F81798u, 0x59F2815Bu, 0x2DCE28D9u, 0x029BFCDBu, 0xCE870B07u, 0x55A06295u, 0xF9DCBBACu, 0x79BE667Eu};
        const uint32_t GY[8] = {0xFB10D4B8u, 0x9C47D08Fu, 0xA6855419u, 0xFD17B448u, 0x0E1108A8u, 0x5DA4FBFCu, 0x26A3C465u, 0x483ADA77u};
        affine_mem m;
        for (int i = 0; i < 8; i++) {
            m.x.w[i] = GX[i];
            m.y.w[i] = GY[i];
        }
        out[p] = m;
        return;
    }
    uint64_t label;
    if (p < nm)
        label = seed + p;
    else if (p < 2 * nm)
        label = seed + n_bits + (p - nm);
    else
        label = kind == 0 ? 1 : seed - 1;
    uint64_t h[8];
    sha512_label(label, h);
    uint8_t bytes[64];
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) bytes[8 * i + j] = (uint8_t)(h[i] >> (56 - 8 * j));
    // to_bytes(from_bytes(digest)): drop leading zero bytes
    int z = 0;
    while (z < 63 && bytes[z] == 0) z++;
    const int len = 64 - z;
    if (z)
        for (int i = 0; i < len; i++) bytes[i] = bytes[z + i];
    out[p] = affine_store(grp_from_bytes(bytes, len));
}

hipError_t launch_gens_derive(int kind, uint64_t seed, uint64_t n_bits, size_t nm, affine_mem* out, hipStream_t st) {
    const size_t np = 2 * nm + 2;
    hipLaunchKernelGGL(k_gens_derive, dim3((unsigned)((np + 63) / 64)), dim3(64), 0, st, kind, seed, n_bits, nm, out);
    return hipGetLastError();
}

}  // namespace bp
