// CPU-side unit-test hooks for the __host__ __device__ arithmetic headers (fp.h, fq.h, ec.h,
// sha256.h).  Built with g++ into bulletproofs_amd/libbpmsm_hosttest.so and driven through ctypes
// by tests/test_host_arith.py (no GPU needed).  Not part of the product ABI.
#include <string.h>

#include "ec.h"
#include "ecmul.h"
#include "fq.h"
#include "hostmath.h"
#include "sha256.h"

using namespace bp;

extern "C" {

void bpt_fp_mul(uint8_t* out, const uint8_t* a, const uint8_t* b) { fp_to_be(out, fp_mul(fp_from_be(a), fp_from_be(b))); }
void bpt_fp_sqr(uint8_t* out, const uint8_t* a) { fp_to_be(out, fp_sqr(fp_from_be(a))); }
void bpt_fp_add(uint8_t* out, const uint8_t* a, const uint8_t* b) { fp_to_be(out, fp_add(fp_from_be(a), fp_from_be(b))); }
void bpt_fp_sub(uint8_t* out, const uint8_t* a, const uint8_t* b) { fp_to_be(out, fp_sub(fp_from_be(a), fp_from_be(b))); }
void bpt_fp_inv(uint8_t* out, const uint8_t* a) { fp_to_be(out, fp_inv(fp_from_be(a))); }
void bpt_fp_inv_var(uint8_t* out, const uint8_t* a) { fp_to_be(out, fp_inv_var(fp_from_be(a))); }
void bpt_fp_sqrt(uint8_t* out, const uint8_t* a) { fp_to_be(out, fp_sqrt_candidate(fp_from_be(a))); }
// linear combinations with one normalisation and the lazy operands (fp.h); inputs are first passed through a multiplication and
// an addition so that they are normalised-but-not-canonical values as the point formulas see them: x = a*b + c, y = b*c + a,
// z = c*a + b.  op: 0 x-y-z, 1 x-y-2z, 2 x-2y, 3 x-4y, 4 3x, 5 4x, 6 (x-y)*z lazy, 7 (2x-y)*z lazy, 8 (2x)*z lazy,
// 9 (-x)*z lazy, 10 (+x)*z lazy
void bpt_fp_lin(uint8_t* out, const uint8_t* a, const uint8_t* b, const uint8_t* c, int op) {
    fp A = fp_from_be(a), B = fp_from_be(b), C = fp_from_be(c);
    fp x = fp_add(fp_mul(A, B), C), y = fp_add(fp_mul(B, C), A), z = fp_add(fp_mul(C, A), B);
    fp r = fp_zero();
    switch (op) {
        case 0: r = fp_sub2(x, y, z); break;
        case 1: r = fp_sub_b_2c(x, y, z); break;
        case 2: r = fp_sub_2b(x, y); break;
        case 3: r = fp_sub_4b(x, y); break;
        case 4: r = fp_mul3(x); break;
        case 5: r = fp_mul4(x); break;
        case 6: r = fp_mul(fp_sub_lazy(x, y), z); break;
        case 7: r = fp_mul(z, fp_2a_sub_lazy(x, y)); break;
        case 8: r = fp_mul(fp_dbl_lazy(x), z); break;
        case 9: r = fp_mul(fp_cneg_lazy(x, true), z); break;
        case 10: r = fp_mul(z, fp_cneg_lazy(x, false)); break;
    }
    fp_to_be(out, r);
}
// zero tests: of raw (possibly slack) limbs, and of the lazily normalised values the point formulas ask about
int bpt_fp_is_zero_limbs(const uint32_t* limbs) {
    fp a;
    for (int i = 0; i < 9; i++) a.v[i] = limbs[i];
    return fp_is_zero(a) ? 1 : 0;
}
// bit 0: a - b == 0, bit 1: a*b - b*a == 0, bit 2: (a + b) - b - a == 0, bit 3: fp_eq(a, b)
int bpt_fp_zero_forms(const uint8_t* a, const uint8_t* b) {
    fp x = fp_from_be(a), y = fp_from_be(b);
    int r = 0;
    if (fp_is_zero(fp_sub(x, y))) r |= 1;
    if (fp_is_zero(fp_sub(fp_mul(x, y), fp_mul(y, x)))) r |= 2;
    if (fp_is_zero(fp_sub(fp_sub(fp_add(x, y), y), x))) r |= 4;
    if (fp_eq(x, y)) r |= 8;
    return r;
}

static fq fq_from_be_raw(const uint8_t* b) { return fq_from_be_reduce(b); }
void bpt_fq_mul(uint8_t* out, const uint8_t* a, const uint8_t* b) { fq_to_be(out, fq_mul(fq_from_be_raw(a), fq_from_be_raw(b))); }
void bpt_fq_add(uint8_t* out, const uint8_t* a, const uint8_t* b) { fq_to_be(out, fq_add(fq_from_be_raw(a), fq_from_be_raw(b))); }
void bpt_fq_sub(uint8_t* out, const uint8_t* a, const uint8_t* b) { fq_to_be(out, fq_sub(fq_from_be_raw(a), fq_from_be_raw(b))); }
void bpt_fq_inv(uint8_t* out, const uint8_t* a) { fq_to_be(out, fq_inv(fq_from_be_raw(a))); }
// GLV split: k1, k2 magnitudes (32 B BE each) and sign flags (bit 0: k1 negative, bit 1: k2 negative)
// reduced-radix Montgomery domain: (a*b), (a+b), (a-b), a^-1 computed through fqm and converted back
void bpt_fqm_mul(uint8_t* out, const uint8_t* a, const uint8_t* b) {
    fq_to_be(out, fqm_to_fq(fqm_mul(fqm_from_fq(fq_from_be_raw(a)), fqm_from_fq(fq_from_be_raw(b)))));
}
void bpt_fqm_add(uint8_t* out, const uint8_t* a, const uint8_t* b) {
    fq_to_be(out, fqm_to_fq(fqm_add(fqm_from_fq(fq_from_be_raw(a)), fqm_from_fq(fq_from_be_raw(b)))));
}
void bpt_fqm_sub(uint8_t* out, const uint8_t* a, const uint8_t* b) {
    fq_to_be(out, fqm_to_fq(fqm_sub(fqm_from_fq(fq_from_be_raw(a)), fqm_from_fq(fq_from_be_raw(b)))));
}
void bpt_fqm_inv_var(uint8_t* out, const uint8_t* a) { fq_to_be(out, fqm_to_fq(fqm_inv_var(fqm_from_fq(fq_from_be_raw(a))))); }
void bpt_fqm_inv(uint8_t* out, const uint8_t* a) { fq_to_be(out, fqm_to_fq(fqm_inv(fqm_from_fq(fq_from_be_raw(a))))); }
// ((a - b) * (a + b) + a) * b - a : chains of lazily reduced values
void bpt_fqm_chain(uint8_t* out, const uint8_t* a, const uint8_t* b) {
    fqm x = fqm_from_fq(fq_from_be_raw(a)), y = fqm_from_fq(fq_from_be_raw(b));
    fqm t = fqm_sub(fqm_mul(fqm_add(fqm_mul(fqm_sub(x, y), fqm_add(x, y)), x), y), x);
    out[32] = fqm_is_zero(t) ? 1 : 0;
    fq_to_be(out, fqm_to_fq(t));
}
int bpt_fq_glv(uint8_t* k1, uint8_t* k2, const uint8_t* k) {
    fq a, b;
    bool n1, n2;
    fq_glv_split(fq_from_be_raw(k), a, n1, b, n2);
    fq_to_be(k1, a);
    fq_to_be(k2, b);
    return (n1 ? 1 : 0) | (n2 ? 2 : 0);
}
// host scalar class of the protocol layer (hostmath.h: 4 x 64-bit Montgomery limbs); op: 0 mul, 1 add, 2 sub, 3 inv(a),
// 4 neg(a), 5 a^(low 64 bits of b), 6 ((a - b) * (a + b) + a) * b - a, 7 a == b (out[31]), 8 inv(a) by exponentiation
void bpt_hostfq(uint8_t* out, const uint8_t* a, const uint8_t* b, int op) {
    Fq x = Fq::from_be(a), y = Fq::from_be(b), r;
    uint64_t e = 0;
    for (int i = 24; i < 32; i++) e = (e << 8) | b[i];
    switch (op) {
        case 0: r = x * y; break;
        case 1: r = x + y; break;
        case 2: r = x - y; break;
        case 3: r = x.inv(); break;
        case 4: r = -x; break;
        case 5: r = x.pow(e); break;
        case 6: r = ((x - y) * (x + y) + x) * y - x; break;
        case 8: r = x.inv_fermat(); break;  // the exponentiation form (inv() runs division steps)
        default: r = Fq(x == y ? 1 : 0); break;
    }
    r.to_be(out);
}
void bpt_fq_reduce(uint8_t* out, const uint8_t* a) { fq_to_be(out, fq_from_be_raw(a)); }

// out = a + b through XYZZ (mixed add), normalised
void bpt_ec_madd(uint8_t* out, const uint8_t* a, const uint8_t* b) {
    xyzz acc = xyzz_from_affine(affine_from_be64(a));
    xyzz_madd(acc, affine_from_be64(b));
    affine_to_be64(out, xyzz_to_affine(acc));
}
// out = (a + a) + b with a full XYZZ add of two non-trivial Z operands, normalised
void bpt_ec_dbl_add(uint8_t* out, const uint8_t* a, const uint8_t* b) {
    xyzz da = xyzz_dbl(xyzz_from_affine(affine_from_be64(a)));
    xyzz db = xyzz_dbl_affine(affine_from_be64(b));
    xyzz_madd(db, affine_neg(affine_from_be64(b)));  // db = b (non-trivial ZZ)
    xyzz_add(da, db);
    affine_to_be64(out, xyzz_to_affine(da));
}
// out = k * a (double-and-add through XYZZ), normalised
void bpt_ec_mul(uint8_t* out, const uint8_t* a, const uint8_t* k) {
    affine p = affine_from_be64(a);
    xyzz acc = xyzz_inf();
    for (int i = 0; i < 256; i++) {
        acc = xyzz_dbl(acc);
        if ((k[i / 8] >> (7 - (i % 8))) & 1) xyzz_madd(acc, p);
    }
    affine_to_be64(out, xyzz_to_affine(acc));
}
// out = k * a through the GLV fixed-window ladder of ecmul.h, normalised
void bpt_ec_mul_glv(uint8_t* out, const uint8_t* a, const uint8_t* k) {
    xyzz tab[ECMUL_NT];
    xyzz r = ec_mul_glv(affine_from_be64(a), fq_from_be_reduce(k), tab);
    affine_to_be64(out, xyzz_to_affine(r));
}
// the four lane-slices of ec_mul_glv_slice summed: must equal k * a
void bpt_ec_mul_glv_sliced(uint8_t* out, const uint8_t* a, const uint8_t* k) {
    xyzz tab[ECMUL_NT];
    xyzz acc = xyzz_inf();
    for (int s = 0; s < 4; s++) {
        xyzz r = ec_mul_glv_slice(affine_from_be64(a), fq_from_be_reduce(k), s, tab);
        xyzz_add(acc, r);
    }
    affine_to_be64(out, xyzz_to_affine(acc));
}
// 2^(k+1) * a: one XYZZ doubling (so ZZ != 1) followed by the Jacobian k-fold doubling chain
void bpt_ec_dbl_n(uint8_t* out, const uint8_t* a, int k) {
    xyzz r = xyzz_dbl(xyzz_from_affine(affine_from_be64(a)));
    affine_to_be64(out, xyzz_to_affine(xyzz_dbl_n(r, k)));
}
// sum_j k_j * a_j for cnt <= 4 points through the one-lane interleaved routine
void bpt_ec_mul_glv_multi(uint8_t* out, const uint8_t* a, const uint8_t* k, int cnt) {
    xyzz tab[4 * ECMUL_SLOTS];
    affine_mem atab[4 * ECMUL_NT];
    alignas(16) int8_t dig[4 * ECMUL_DIGB];
    xyzz r = ec_mul_glv_multi<4>([&](int j) { return affine_from_be64(a + 64 * j); }, [&](int j) { return fq_from_be_reduce(k + 32 * j); }, cnt,
                                 tab, atab, dig);
    affine_to_be64(out, xyzz_to_affine(r));
}
int bpt_ec_decompress(uint8_t* out, const uint8_t* in33) {
    affine p;
    if (!affine_decompress(p, in33)) return -1;
    affine_to_be64(out, p);
    return 0;
}
int bpt_host_decompress(uint8_t* out, const uint8_t* in33) { return host_decompress(in33, out) ? 1 : 0; }
void bpt_fp64_mul(uint8_t* out, const uint8_t* a, const uint8_t* b) { (Fp64::from_be(a) * Fp64::from_be(b)).to_be(out); }
int bpt_ec_on_curve(const uint8_t* a) { return affine_on_curve(affine_from_be64(a)) ? 1 : 0; }

void bpt_sha256(uint8_t* out, const uint8_t* data, uint32_t len) {
    sha256_ctx c;
    sha256_init(c);
    sha256_update(c, data, len);
    sha256_final(c, out);
}
}
