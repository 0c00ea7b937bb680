// Sanitizer run of the host-side arithmetic (fp.h, fq.h, ec.h, ecmul.h, hostmath.h, sha256.h as compiled for the CPU):
// a standalone main() around host_selftest.cpp's hooks, built by `make san` with -fsanitize=address,undefined and run by
// tests/test_host_sanitizers.py.  Self-consistency identities on seeded random inputs; any out-of-bounds access, signed
// overflow or misaligned / invalid shift in the headers aborts the run (-fno-sanitize-recover=all).
#include <stdio.h>
#include <string.h>

#include "host_selftest.cpp"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}
static void rnd32(uint8_t* b, bool below_p) {
    for (int i = 0; i < 32; i++) b[i] = (uint8_t)rnd();
    if (below_p) b[0] &= 0x7F;  // < 2^255 < p, q
}
static int fails = 0;
static void expect(bool ok, const char* what) {
    if (!ok) {
        fprintf(stderr, "FAIL: %s\n", what);
        fails++;
    }
}

int main() {
    static const uint8_t G[64] = {0x79, 0xBE, 0x66, 0x7E, 0xF9, 0xDC, 0xBB, 0xAC, 0x55, 0xA0, 0x62, 0x95, 0xCE, 0x87, 0x0B, 0x07,
                                  0x02, 0x9B, 0xFC, 0xDB, 0x2D, 0xCE, 0x28, 0xD9, 0x59, 0xF2, 0x81, 0x5B, 0x16, 0xF8, 0x17, 0x98,
                                  0x48, 0x3A, 0xDA, 0x77, 0x26, 0xA3, 0xC4, 0x65, 0x5D, 0xA4, 0xFB, 0xFC, 0x0E, 0x11, 0x08, 0xA8,
                                  0xFD, 0x17, 0xB4, 0x48, 0xA6, 0x85, 0x54, 0x19, 0x9C, 0x47, 0xD0, 0x8F, 0xFB, 0x10, 0xD4, 0xB8};
    uint8_t a[32], b[32], c[32], d[32], e[32], one[32] = {0};
    one[31] = 1;
    expect(bpt_ec_on_curve(G) == 1, "G on curve");
    // one-normalisation linear combinations and lazy multiplier operands, limb ranges and column sums checked (BP_CHECK_BOUNDS):
    // all-ones patterns drive every limb to its maximum
    for (int it = 0; it < 2000; it++) {
        uint8_t x[32], y[32], z[32], o[32];
        rnd32(x, false);
        rnd32(y, false);
        rnd32(z, false);
        if (it % 7 == 0) memset(x, 0xFF, 32);
        if (it % 11 == 0) memset(y, 0xFF, 32);
        if (it % 13 == 0) memset(z, 0xFF, 32);
        if (it % 17 == 0) memset(y, 0, 32);
        for (int op = 0; op <= 10; op++) bpt_fp_lin(o, x, y, z, op);
    }
    for (int it = 0; it < 300; it++) {
        rnd32(a, true);
        rnd32(b, true);
        a[31] |= 1;
        b[31] |= 1;
        // base field: (a*b)*b^-1 == a with both inversions; sqrt(a^2)^2 == a^2
        bpt_fp_mul(c, a, b);
        bpt_fp_inv_var(d, b);
        bpt_fp_mul(e, c, d);
        expect(memcmp(e, a, 32) == 0, "fp: (a b) b^-1 == a (inv_var)");
        if (it < 20) {
            bpt_fp_inv(e, b);
            expect(memcmp(e, d, 32) == 0, "fp: Fermat inverse == division-step inverse");
        }
        bpt_fp_sqr(c, a);
        bpt_fp_sqrt(d, c);
        bpt_fp_sqr(e, d);
        expect(memcmp(e, c, 32) == 0, "fp: sqrt(a^2)^2 == a^2");
        bpt_fp_add(c, a, b);
        bpt_fp_sub(d, c, b);
        expect(memcmp(d, a, 32) == 0, "fp: (a + b) - b == a");
        // scalar field: plain, Montgomery 9x29 and host class agree
        bpt_fq_mul(c, a, b);
        bpt_fqm_mul(d, a, b);
        bpt_hostfq(e, a, b, 0);
        expect(memcmp(c, d, 32) == 0 && memcmp(c, e, 32) == 0, "fq: three multipliers agree");
        bpt_fqm_inv_var(c, b);
        bpt_hostfq(d, b, b, 3);
        expect(memcmp(c, d, 32) == 0, "fq: division-step inverse == host inverse");
        bpt_fq_mul(e, c, b);
        expect(memcmp(e, one, 32) == 0, "fq: b b^-1 == 1");
        uint8_t ch[33];
        bpt_fqm_chain(ch, a, b);
        bpt_hostfq(e, a, b, 6);
        expect(memcmp(ch, e, 32) == 0, "fq: expression chain, Montgomery 9x29 == host class");
        uint8_t k1[32], k2[32];
        (void)bpt_fq_glv(k1, k2, a);
        expect(k1[0] == 0 && k2[0] == 0 && k1[15] < 4 && k2[15] < 4, "GLV halves are ~128 bits");
    }
    // group: k*G three ways; (k1 + k2) G == k1 G + k2 G; decompression round trip
    for (int it = 0; it < 12; it++) {
        uint8_t k[32], l[32], s[32], P[64], Q[64], R[64], S[64];
        rnd32(k, true);
        rnd32(l, true);
        bpt_ec_mul(P, G, k);
        bpt_ec_mul_glv(Q, G, k);
        bpt_ec_mul_glv_sliced(R, G, k);
        expect(memcmp(P, Q, 64) == 0 && memcmp(P, R, 64) == 0, "ec: ladder == GLV windows == four slices");
        expect(bpt_ec_on_curve(P) == 1, "ec: k G on curve");
        bpt_ec_mul_glv(Q, G, l);
        bpt_ec_madd(R, P, Q);
        bpt_hostfq(s, k, l, 1);
        bpt_ec_mul_glv(S, G, s);
        expect(memcmp(R, S, 64) == 0, "ec: k G + l G == (k + l) G");
        uint8_t pts[4 * 64], ks[4 * 32];
        memcpy(pts, P, 64);
        memcpy(pts + 64, Q, 64);
        memcpy(pts + 128, G, 64);
        memcpy(pts + 192, R, 64);
        for (int j = 0; j < 4; j++) rnd32(ks + 32 * j, true);
        bpt_ec_mul_glv_multi(S, pts, ks, 1 + it % 4);
        uint8_t acc[64], t[64];
        bpt_ec_mul_glv(acc, pts, ks);
        for (int j = 1; j < 1 + it % 4; j++) {
            bpt_ec_mul_glv(t, pts + 64 * j, ks + 32 * j);
            uint8_t u[64];
            bpt_ec_madd(u, acc, t);
            memcpy(acc, u, 64);
        }
        expect(memcmp(acc, S, 64) == 0, "ec: interleaved multi-point ladder == sum of single ladders");
        uint8_t c33[33], back[64];
        c33[0] = (P[63] & 1) ? 3 : 2;
        memcpy(c33 + 1, P, 32);
        expect(bpt_ec_decompress(back, c33) == 0 && memcmp(back, P, 64) == 0, "ec: device-code decompression round trip");
        expect(bpt_host_decompress(back, c33) == 1 && memcmp(back, P, 64) == 0, "ec: host decompression round trip");
        bpt_ec_dbl_n(Q, P, 5);
        uint8_t k64[32] = {0};
        k64[31] = 64;
        bpt_ec_mul(R, P, k64);
        expect(memcmp(Q, R, 64) == 0, "ec: doubling chain == 64 P");
    }
    // SHA-256 known answer ("abc") and a two-block message
    {
        static const uint8_t want[32] = {0xba, 0x78, 0x16, 0xbf, 0x8f, 0x01, 0xcf, 0xea, 0x41, 0x41, 0x40, 0xde, 0x5d, 0xae, 0x22, 0x23,
                                         0xb0, 0x03, 0x61, 0xa3, 0x96, 0x17, 0x7a, 0x9c, 0xb4, 0x10, 0xff, 0x61, 0xf2, 0x00, 0x15, 0xad};
        uint8_t out[32];
        bpt_sha256(out, (const uint8_t*)"abc", 3);
        expect(memcmp(out, want, 32) == 0, "sha256(abc)");
        uint8_t msg[200];
        for (int i = 0; i < 200; i++) msg[i] = (uint8_t)i;
        for (int len = 50; len <= 200; len += 13) bpt_sha256(out, msg, (uint32_t)len);
    }
    // generator recipe (hostmath.h label_point: SHA-512 + try-and-double): points come out on the curve
    {
        uint8_t P[64];
        for (uint64_t lab = 1; lab < 40; lab++) {
            label_point(u64_min_be(0x4B5A656Eull + lab), P);
            expect(bpt_ec_on_curve(P) == 1, "label_point on curve");
        }
    }
    if (!fails) printf("host_selftest_san OK\n");
    return fails ? 1 : 0;
}
