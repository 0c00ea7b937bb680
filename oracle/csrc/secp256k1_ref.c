/* CPU oracle, plain C — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Restates the group arithmetic that curv-kzen 0.10 (external, Cargo.toml:22) performs behind the
 * reference's operator calls:
 *   `&Point * &Scalar`  (e.g. /root/reference/src/proofs/range_proof.rs:773)  -> ref_point_mul
 *   `Point + Point`     (range_proof.rs:774)                                  -> ref_point_add
 *   the fold idiom `(0..n).map(|i| &p[i] * &s[i]).fold(init, |acc,x| acc + x)`
 *                       (range_proof.rs:770-774, inner_product.rs:336-338)    -> ref_msm_naive
 *   `hi_tag[i] = &h_vec[i] * &yi_inv[i]` (range_proof.rs:231-233)             -> ref_scalar_mul_batch
 * Each term is a full scalar multiplication followed by normalisation to a canonical affine
 * point and an affine addition — the reference's cost structure, which is why this file is also
 * the `cpu_baseline` ("port") timed by bench.py.
 *
 * Deliberately a different implementation from the product (4x64-bit limbs + unsigned __int128,
 * Jacobian coordinates, Fermat inversion) so that agreement is evidence.
 * Boundary encoding: 64 bytes x||y big-endian; 64 zero bytes = infinity; scalars 32 bytes BE.
 */
#include <stdint.h>
#include <string.h>
#include <stddef.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } fe;            /* little-endian limbs, value < p */
typedef struct { fe X, Y, Z; int inf; } jac;     /* Jacobian; inf flag explicit     */

static const fe FE_P = {{0xFFFFFFFEFFFFFC2FULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL}};
#define PC 0x1000003D1ULL /* 2^256 - p */

static int fe_is_zero(const fe *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static int fe_eq(const fe *a, const fe *b) { return memcmp(a, b, sizeof(fe)) == 0; }
static int fe_geq_p(const fe *a) {
    for (int i = 3; i >= 0; i--) {
        if (a->v[i] > FE_P.v[i]) return 1;
        if (a->v[i] < FE_P.v[i]) return 0;
    }
    return 1;
}
static void fe_sub_p(fe *a) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)a->v[i] - FE_P.v[i] - br;
        a->v[i] = (uint64_t)t;
        br = (t >> 64) & 1;
    }
}
static void fe_add(fe *r, const fe *a, const fe *b) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a->v[i] + b->v[i]; r->v[i] = (uint64_t)c; c >>= 64; }
    if (c) { /* wrapped 2^256: add 2^256 mod p */
        u128 t = (u128)r->v[0] + PC; r->v[0] = (uint64_t)t; t >>= 64;
        for (int i = 1; i < 4 && t; i++) { t += r->v[i]; r->v[i] = (uint64_t)t; t >>= 64; }
    }
    if (fe_geq_p(r)) fe_sub_p(r);
}
static void fe_neg(fe *r, const fe *a) {
    if (fe_is_zero(a)) { *r = *a; return; }
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)FE_P.v[i] - a->v[i] - br;
        r->v[i] = (uint64_t)t;
        br = (t >> 64) & 1;
    }
}
static void fe_sub(fe *r, const fe *a, const fe *b) { fe nb; fe_neg(&nb, b); fe_add(r, a, &nb); }
static void fe_mul(fe *r, const fe *a, const fe *b) {
    uint64_t t[8] = {0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a->v[i] * b->v[j] + t[i + j];
            t[i + j] = (uint64_t)c;
            c >>= 64;
        }
        t[i + 4] = (uint64_t)c;
    }
    /* fold hi*PC into lo, twice */
    uint64_t lo[5];
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)t[4 + i] * PC + t[i]; lo[i] = (uint64_t)c; c >>= 64; }
    lo[4] = (uint64_t)c;
    c = (u128)lo[4] * PC;
    for (int i = 0; i < 4; i++) { c += lo[i]; r->v[i] = (uint64_t)c; c >>= 64; }
    if (c) { /* one more wrap */
        u128 d = (u128)r->v[0] + PC; r->v[0] = (uint64_t)d; d >>= 64;
        for (int i = 1; i < 4 && d; i++) { d += r->v[i]; r->v[i] = (uint64_t)d; d >>= 64; }
    }
    if (fe_geq_p(r)) fe_sub_p(r);
}
static void fe_sqr(fe *r, const fe *a) { fe_mul(r, a, a); }
static void fe_pow(fe *r, const fe *a, const fe *e) {
    fe acc = {{1, 0, 0, 0}};
    for (int i = 255; i >= 0; i--) {
        fe_sqr(&acc, &acc);
        if ((e->v[i / 64] >> (i % 64)) & 1) fe_mul(&acc, &acc, a);
    }
    *r = acc;
}
static void fe_inv(fe *r, const fe *a) {
    fe e = FE_P; e.v[0] -= 2; fe_pow(r, a, &e);
}
static void fe_from_be(fe *r, const uint8_t *b) {
    for (int i = 0; i < 4; i++) {
        uint64_t w = 0;
        for (int j = 0; j < 8; j++) w = (w << 8) | b[(3 - i) * 8 + j];
        r->v[i] = w;
    }
}
static void fe_to_be(uint8_t *b, const fe *a) {
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) b[(3 - i) * 8 + j] = (uint8_t)(a->v[i] >> (56 - 8 * j));
}

static void jac_dbl(jac *r, const jac *p) {
    if (p->inf || fe_is_zero(&p->Y)) { r->inf = 1; return; }
    fe YY, S, M, t, X3, Y3, Z3, YYYY;
    fe_sqr(&YY, &p->Y);
    fe_mul(&S, &p->X, &YY); fe_add(&S, &S, &S); fe_add(&S, &S, &S);        /* 4XY^2 */
    fe_sqr(&M, &p->X); fe_add(&t, &M, &M); fe_add(&M, &t, &M);             /* 3X^2  */
    fe_sqr(&X3, &M); fe_sub(&X3, &X3, &S); fe_sub(&X3, &X3, &S);
    fe_sqr(&YYYY, &YY); fe_add(&YYYY, &YYYY, &YYYY); fe_add(&YYYY, &YYYY, &YYYY); fe_add(&YYYY, &YYYY, &YYYY);
    fe_sub(&t, &S, &X3); fe_mul(&Y3, &M, &t); fe_sub(&Y3, &Y3, &YYYY);
    fe_mul(&Z3, &p->Y, &p->Z); fe_add(&Z3, &Z3, &Z3);
    r->X = X3; r->Y = Y3; r->Z = Z3; r->inf = 0;
}
static void jac_add_affine(jac *r, const jac *p, const fe *x2, const fe *y2) {
    if (p->inf) { r->X = *x2; r->Y = *y2; r->Z = (fe){{1, 0, 0, 0}}; r->inf = 0; return; }
    fe ZZ, U2, S2, H, R, HH, HHH, V, X3, Y3, Z3, t;
    fe_sqr(&ZZ, &p->Z);
    fe_mul(&U2, x2, &ZZ);
    fe_mul(&S2, y2, &p->Z); fe_mul(&S2, &S2, &ZZ);
    fe_sub(&H, &U2, &p->X);
    fe_sub(&R, &S2, &p->Y);
    if (fe_is_zero(&H)) {
        if (fe_is_zero(&R)) { jac_dbl(r, p); return; }
        r->inf = 1; return;
    }
    fe_sqr(&HH, &H); fe_mul(&HHH, &H, &HH); fe_mul(&V, &p->X, &HH);
    fe_sqr(&X3, &R); fe_sub(&X3, &X3, &HHH); fe_sub(&X3, &X3, &V); fe_sub(&X3, &X3, &V);
    fe_sub(&t, &V, &X3); fe_mul(&Y3, &R, &t); fe_mul(&t, &p->Y, &HHH); fe_sub(&Y3, &Y3, &t);
    fe_mul(&Z3, &p->Z, &H);
    r->X = X3; r->Y = Y3; r->Z = Z3; r->inf = 0;
}
typedef struct { fe x, y; int inf; } aff;
static void jac_to_aff(aff *r, const jac *p) {
    if (p->inf || fe_is_zero(&p->Z)) { r->inf = 1; memset(&r->x, 0, sizeof(fe)); memset(&r->y, 0, sizeof(fe)); return; }
    fe zi, zi2, zi3;
    fe_inv(&zi, &p->Z); fe_sqr(&zi2, &zi); fe_mul(&zi3, &zi2, &zi);
    fe_mul(&r->x, &p->X, &zi2); fe_mul(&r->y, &p->Y, &zi3); r->inf = 0;
}
static void aff_from_bytes(aff *r, const uint8_t *b) {
    int z = 1;
    for (int i = 0; i < 64; i++) if (b[i]) { z = 0; break; }
    r->inf = z;
    fe_from_be(&r->x, b); fe_from_be(&r->y, b + 32);
}
static void aff_to_bytes(uint8_t *b, const aff *a) {
    if (a->inf) { memset(b, 0, 64); return; }
    fe_to_be(b, &a->x); fe_to_be(b + 32, &a->y);
}
/* k * P, MSB-first double-and-add (variable time; oracle only), result normalised to affine. */
static void aff_mul(aff *r, const aff *p, const uint8_t k[32]) {
    jac acc; acc.inf = 1;
    if (!p->inf) {
        for (int i = 0; i < 256; i++) {
            jac t; jac_dbl(&t, &acc); acc = t;
            if ((k[i / 8] >> (7 - (i % 8))) & 1) { jac_add_affine(&t, &acc, &p->x, &p->y); acc = t; }
        }
    }
    jac_to_aff(r, &acc);
}
static void aff_add(aff *r, const aff *a, const aff *b) {
    if (a->inf) { *r = *b; return; }
    if (b->inf) { *r = *a; return; }
    jac ja = { a->x, a->y, {{1, 0, 0, 0}}, 0 }, t;
    jac_add_affine(&t, &ja, &b->x, &b->y);
    jac_to_aff(r, &t);
}

void ref_point_mul(uint8_t out[64], const uint8_t in[64], const uint8_t k[32]) {
    aff p, r; aff_from_bytes(&p, in); aff_mul(&r, &p, k); aff_to_bytes(out, &r);
}
void ref_point_add(uint8_t out[64], const uint8_t a[64], const uint8_t b[64]) {
    aff pa, pb, r; aff_from_bytes(&pa, a); aff_from_bytes(&pb, b); aff_add(&r, &pa, &pb); aff_to_bytes(out, &r);
}
void ref_msm_naive(uint8_t out[64], const uint8_t *points, const uint8_t *scalars, size_t n, const uint8_t init[64]) {
    aff acc; aff_from_bytes(&acc, init);
    for (size_t i = 0; i < n; i++) {
        aff p, t, s;
        aff_from_bytes(&p, points + 64 * i);
        aff_mul(&t, &p, scalars + 32 * i);      /* &points[i] * &scalars[i]  */
        aff_add(&s, &acc, &t);                  /* acc + x                   */
        acc = s;
    }
    aff_to_bytes(out, &acc);
}
void ref_scalar_mul_batch(uint8_t *out, const uint8_t *points, const uint8_t *scalars, size_t n) {
    for (size_t i = 0; i < n; i++) ref_point_mul(out + 64 * i, points + 64 * i, scalars + 32 * i);
}
/* 33-byte SEC1 -> 64-byte affine; returns 0 on success, -1 if not a curve point. */
int ref_decompress(uint8_t out[64], const uint8_t in[33]) {
    if (in[0] != 2 && in[0] != 3) return -1;
    fe x; fe_from_be(&x, in + 1);
    if (fe_geq_p(&x)) return -1;
    fe y2, y, t, e = {{0xFFFFFFFFBFFFFF0CULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0x3FFFFFFFFFFFFFFFULL}}; /* (p+1)/4 */
    fe seven = {{7, 0, 0, 0}};
    fe_sqr(&y2, &x); fe_mul(&y2, &y2, &x); fe_add(&y2, &y2, &seven);
    fe_pow(&y, &y2, &e);
    fe_sqr(&t, &y);
    if (!fe_eq(&t, &y2)) return -1;
    if ((y.v[0] & 1) != (uint64_t)(in[0] & 1)) fe_neg(&y, &y);
    fe_to_be(out, &x); fe_to_be(out + 32, &y);
    return 0;
}
