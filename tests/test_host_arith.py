"""The product's __host__ __device__ arithmetic headers (fp.h, fq.h, ec.h, sha256.h), compiled for
the CPU into libbpmsm_hosttest.so, against Python big ints / hashlib / the oracle."""
import ctypes
import hashlib
import os
import random
import subprocess

import pytest

from oracle import ec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, Q = ec.P, ec.Q


@pytest.fixture(scope="module")
def L():
    path = os.path.join(ROOT, "bulletproofs_amd", "libbpmsm_hosttest.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "bulletproofs_amd", "csrc"), "../libbpmsm_hosttest.so"])
    return ctypes.CDLL(path)


def b32(x):
    return x.to_bytes(32, "big")


def call2(f, a, b):
    o = ctypes.create_string_buffer(32)
    f(o, b32(a), b32(b))
    return int.from_bytes(o.raw, "big")


def call1(f, a):
    o = ctypes.create_string_buffer(32)
    f(o, b32(a))
    return int.from_bytes(o.raw, "big")


def test_fp(L):
    rng = random.Random(7)
    edge = [0, 1, 2, P - 1, P - 2, 2**32, 2**32 + 977, 2**255, P - 2**32 - 977, (1 << 224) - 1, P - 977]
    vals = edge + [rng.randrange(P) for _ in range(120)]
    for a in vals:
        for b in rng.sample(vals, 8) + edge:
            assert call2(L.bpt_fp_mul, a, b) == a * b % P
            assert call2(L.bpt_fp_add, a, b) == (a + b) % P
            assert call2(L.bpt_fp_sub, a, b) == (a - b) % P
        assert call1(L.bpt_fp_sqr, a) == a * a % P
    for a in vals[:40]:
        if a:
            assert call1(L.bpt_fp_inv, a) == pow(a, -1, P)
        assert call1(L.bpt_fp_sqrt, a) == pow(a, (P + 1) // 4, P)


def test_fp_linear_combinations_and_lazy_operands(L):
    """fp.h's one-normalisation linear combinations and the lazy (unnormalised) multiplier operands, on extreme and random inputs."""
    rng = random.Random(13)
    edge = [0, 1, P - 1, P - 2, 2**255, P - 2**32 - 977, (1 << 256) - 1 - (1 << 32), 2**32 + 977, (1 << 232) - 1, P >> 1]
    vals = edge + [rng.randrange(P) for _ in range(60)]
    want = [lambda x, y, z: x - y - z, lambda x, y, z: x - y - 2 * z, lambda x, y, z: x - 2 * y, lambda x, y, z: x - 4 * y,
            lambda x, y, z: 3 * x, lambda x, y, z: 4 * x, lambda x, y, z: (x - y) * z, lambda x, y, z: (2 * x - y) * z,
            lambda x, y, z: 2 * x * z, lambda x, y, z: -x * z, lambda x, y, z: x * z]
    for a in vals:
        for _ in range(6):
            b, c = rng.choice(vals), rng.choice(vals)
            x, y, z = (a * b + c) % P, (b * c + a) % P, (c * a + b) % P
            for op, f in enumerate(want):
                o = ctypes.create_string_buffer(32)
                L.bpt_fp_lin(o, b32(a % (1 << 256)), b32(b), b32(c), op)
                assert int.from_bytes(o.raw, "big") == f(x, y, z) % P, (op, a, b, c)


def test_fp_is_zero_prefilter(L):
    """fp_is_zero answers from limb 0's low 29 bits unless they are those of 0 or p: exact on slack limb forms of 0 and p, on
    values that share p's or 0's low bits, and on the differences the point formulas test."""
    import ctypes
    rng = random.Random(11)
    M = (1 << 29) - 1

    def limbs_of(v):
        return [(v >> (29 * i)) & M for i in range(8)] + [v >> 232]

    def call(l):
        return L.bpt_fp_is_zero_limbs((ctypes.c_uint32 * 9)(*l))

    assert call(limbs_of(0)) == 1 and call(limbs_of(P)) == 1
    # slack forms of p and 0: move a carry between neighbouring limbs (value unchanged, limbs stay below 2^32)
    for base in (P, 0):
        for _ in range(200):
            l = limbs_of(base)
            for _ in range(rng.randrange(1, 4)):
                i = rng.randrange(1, 8)
                if l[i + 1] >= 1 and l[i] + (1 << 29) < (1 << 32):
                    l[i + 1] -= 1
                    l[i] += 1 << 29
            assert sum(x << (29 * i) for i, x in enumerate(l)) == base
            assert call(l) == 1, l
    # non-zero values whose low 29 bits are those of p or of 0
    for low in (P & M, 0):
        for _ in range(300):
            v = (rng.randrange(1, 1 << 227) << 29) | low
            if v % P:
                assert call(limbs_of(v % (1 << 256))) == (1 if (v % (1 << 256)) % P == 0 else 0)
    for _ in range(300):
        assert call(limbs_of(rng.randrange(1, P))) == 0
    # differences as xyzz_madd / xyzz_add form them
    for _ in range(300):
        a = rng.randrange(P)
        assert L.bpt_fp_zero_forms(b32(a), b32(a)) == 15
        b = (a + rng.choice([1, P - 1, 1 << 29, P - (1 << 29), rng.randrange(1, P)])) % P
        assert L.bpt_fp_zero_forms(b32(a), b32(b)) == 6  # a != b; the two identities still hold


def test_fp_inv_var(L):
    """fp_inv_var (Bernstein-Yang division steps, used by the single-lane tails of every MSM) against Python's pow and
    the Fermat chain: edge values, sparse values, small values, values near p, 20000 random ones."""
    rng = random.Random(70)
    vals = [0, 1, 2, 3, P - 1, P - 2, P - 3, 2**32, 2**32 + 977, 2**32 + 976, 2**255, P - 2**32 - 977, (1 << 224) - 1, P - 977, 2**30, 2**30 - 1,
            2**60, 2**240, 2**240 - 1, (P + 1) // 2, (P - 1) // 2]
    vals += [1 << k for k in range(0, 256, 7)] + [P - (1 << k) for k in range(0, 250, 11)]
    vals += [rng.randrange(1 << k) for k in range(1, 256, 3)]
    vals += [rng.randrange(P) for _ in range(20000)]
    for a in vals:
        want = pow(a, -1, P) if a else 0
        assert call1(L.bpt_fp_inv_var, a) == want, hex(a)
    for a in vals[:60]:
        assert call1(L.bpt_fp_inv_var, a) == call1(L.bpt_fp_inv, a)


def test_fqm_inv_var(L):
    """fqm_inv_var (division steps mod the group order q, Montgomery in and out: the batch verifiers' per-proof
    inversion) against Python's pow and the exponentiation chain: edge, sparse, small and 20000 random values."""
    rng = random.Random(72)
    vals = [0, 1, 2, 3, Q - 1, Q - 2, Q - 3, 2**255 % Q, 2**128, 2**29 - 1, 2**30, 2**30 - 1, 2**232, (Q + 1) // 2, (Q - 1) // 2, 2**261 % Q,
            pow(2, -261, Q)]
    vals += [1 << k for k in range(0, 256, 7)] + [Q - (1 << k) for k in range(0, 250, 11)]
    vals += [rng.randrange(1 << k) for k in range(1, 256, 3)]
    vals += [rng.randrange(Q) for _ in range(20000)]
    for a in vals:
        want = pow(a, -1, Q) if a else 0
        assert call1(L.bpt_fqm_inv_var, a) == want, hex(a)
    for a in vals[1:60]:
        assert call1(L.bpt_fqm_inv_var, a) == call1(L.bpt_fqm_inv, a)


def test_host_fp64_and_decompress(L):
    """The host-only 4x64-limb base field (hostmath.h Fp64: proof-point decompression of the single-proof verifiers)
    against Python ints, and host_decompress against the portable decompression and the oracle on residues,
    non-residues, both parities, x >= p and bad prefixes."""
    rng = random.Random(71)
    edge = [0, 1, 2, P - 1, P - 2, 2**32 + 977, 2**32 + 976, 2**255, (1 << 256) - 1 - 2**32 - 977 - 1, 2**64 - 1, 2**128 - 1, 2**192 - 1, P - 977]
    vals = edge + [rng.randrange(P) for _ in range(300)]
    for a in vals:
        for b in rng.sample(vals, 6) + edge:
            assert call2(L.bpt_fp64_mul, a, b) == a * b % P, (hex(a), hex(b))
    xs = [0, 1, 2, 3, 5, P - 1, P - 2, P, P + 1, (1 << 256) - 1, ec.G[0]] + [rng.randrange(P) for _ in range(400)]
    # x with x^3 + 7 close to p (the renormalisation branch)
    xs += [x for x in range(1, 200)]
    for x in xs:
        for pre in (2, 3, 4, 0):
            enc = bytes([pre]) + (x % (1 << 256)).to_bytes(32, "big")
            o1 = ctypes.create_string_buffer(64)
            o2 = ctypes.create_string_buffer(64)
            r1 = L.bpt_host_decompress(o1, enc)
            r2 = L.bpt_ec_decompress(o2, enc)
            assert bool(r1) == (r2 == 0), (pre, hex(x))
            if r1:
                assert o1.raw == o2.raw, (pre, hex(x))
                y = int.from_bytes(o1.raw[32:], "big")
                assert (y * y - x * x * x - 7) % P == 0 and (y & 1) == (pre & 1)


def test_fq(L):
    rng = random.Random(8)
    edge = [0, 1, 2, Q - 1, Q - 2, 2**255, 2**128]
    vals = edge + [rng.randrange(Q) for _ in range(120)]
    for a in vals:
        for b in rng.sample(vals, 8) + edge:
            assert call2(L.bpt_fq_mul, a, b) == a * b % Q
            assert call2(L.bpt_fq_add, a, b) == (a + b) % Q
            assert call2(L.bpt_fq_sub, a, b) == (a - b) % Q
    for a in vals[:30]:
        if a:
            assert call1(L.bpt_fq_inv, a) == pow(a, -1, Q)
    # reduced-radix Montgomery domain (fqm) used by the verifier's scalar kernels
    edge = [0, 1, 2, Q - 1, Q - 2, (Q - 1) // 2, 2**255 % Q, 2**128, 2**29 - 1, 2**232]
    vals = edge + [rng.randrange(Q) for _ in range(40)]
    for a in vals:
        for b in vals[:14]:
            assert call2(L.bpt_fqm_mul, a, b) == a * b % Q
            assert call2(L.bpt_fqm_add, a, b) == (a + b) % Q
            assert call2(L.bpt_fqm_sub, a, b) == (a - b) % Q
            o = ctypes.create_string_buffer(33)
            L.bpt_fqm_chain(o, b32(a), b32(b))
            want = (((a - b) * (a + b) + a) * b - a) % Q
            assert int.from_bytes(o.raw[:32], "big") == want and o.raw[32] == (1 if want == 0 else 0)
        if a:
            assert call1(L.bpt_fqm_inv, a) == pow(a, -1, Q)
    assert call1(L.bpt_fq_reduce, 2**256 - 1) == (2**256 - 1) % Q
    assert call1(L.bpt_fq_reduce, Q) == 0


def test_host_scalar_class(L):
    """hostmath.h `Fq` (4 x 64-bit Montgomery limbs: the protocol layer's scalars) against Python ints."""
    rng = random.Random(18)
    edge = [0, 1, 2, Q - 1, Q - 2, (Q - 1) // 2, 2**255, 2**128, 2**64 - 1, 2**64, 2**192 - 1, 2**256 - 1, Q, Q + 1]
    vals = edge + [rng.randrange(2**256) for _ in range(80)]

    def hf(a, b, op):
        o = ctypes.create_string_buffer(32)
        L.bpt_hostfq(o, b32(a), b32(b), op)
        return int.from_bytes(o.raw, "big")

    for a in vals:
        for b in rng.sample(vals, 6) + edge[:6]:
            assert hf(a, b, 0) == a * b % Q
            assert hf(a, b, 1) == (a + b) % Q
            assert hf(a, b, 2) == (a - b) % Q
            assert hf(a, b, 6) == (((a - b) * (a + b) + a) * b - a) % Q
            assert hf(a, b, 7) == (1 if (a - b) % Q == 0 else 0)
        assert hf(a, 0, 4) == (-a) % Q
        assert hf(a, 65537, 5) == pow(a, 65537, Q)
        assert hf(a, 2**64 - 1, 5) == pow(a, 2**64 - 1, Q)
        if a % Q:
            assert hf(a, 0, 3) == pow(a, -1, Q)  # division steps (the protocol layer's inverse)
            assert hf(a, 0, 8) == pow(a, -1, Q)  # the exponentiation it replaced
        else:
            assert hf(a, 0, 3) == 0
    for _ in range(3000):  # the division-step inverse on many more values (variable number of rounds)
        a = rng.randrange(1, Q)
        assert hf(a, 0, 3) == pow(a, -1, Q)


def test_ec(L, oracle_c):
    rng = random.Random(9)

    def p2(f, a, b):
        o = ctypes.create_string_buffer(64)
        f(o, ec.enc64(a), b if isinstance(b, bytes) else ec.enc64(b))
        return ec.dec64(o.raw)

    pts = [oracle_c.mul(ec.G, rng.randrange(1, Q)) for _ in range(10)]
    for a in pts:
        for b in pts[:4] + [a, ec.neg(a), None]:
            assert p2(L.bpt_ec_madd, a, b) == oracle_c.add(a, b)
            assert p2(L.bpt_ec_dbl_add, a, b) == oracle_c.add(oracle_c.add(a, a), b)
        k = rng.randrange(Q)
        assert p2(L.bpt_ec_mul, a, b32(k)) == oracle_c.mul(a, k)
        o = ctypes.create_string_buffer(64)
        assert L.bpt_ec_decompress(o, ec.compress(a)) == 0 and ec.dec64(o.raw) == a
        assert L.bpt_ec_on_curve(ec.enc64(a)) == 1
    for a in pts[:3]:
        for k in (0, 1, 2, 3, 5, 15, 40):
            o = ctypes.create_string_buffer(64)
            L.bpt_ec_dbl_n(o, ec.enc64(a), k)
            assert ec.dec64(o.raw) == oracle_c.mul(a, 1 << (k + 1)), k
    for cnt in (1, 2, 3, 4):
        ps = [pts[i] for i in range(cnt)]
        if cnt == 3:
            ps[1] = None  # infinity among the points
        if cnt == 4:
            ps[3] = ec.neg(ps[0])
        ks = [rng.randrange(Q) for _ in range(cnt)]
        if cnt == 4:
            ks[3] = ks[0]  # k*P + k*(-P) cancels inside the shared accumulator
        o = ctypes.create_string_buffer(64)
        L.bpt_ec_mul_glv_multi(o, b"".join(ec.enc64(p) for p in ps), b"".join(b32(k) for k in ks), cnt)
        assert ec.dec64(o.raw) == oracle_c.msm_naive(ps, ks, None), cnt
    # the four lane-slices of the latency-path ladder (ec_mul_glv_slice) add up to k * P
    for a in pts[:5]:
        for k in [0, 1, 2, Q - 1, Q - 2, 1 << 68, (1 << 68) - 1, 1 << 128, (1 << 129) - 1, 15, 16, 8, 9] + [rng.randrange(Q) for _ in range(12)]:
            assert p2(L.bpt_ec_mul_glv_sliced, a, b32(k)) == oracle_c.mul(a, k), hex(k)
            assert p2(L.bpt_ec_mul_glv, a, b32(k)) == oracle_c.mul(a, k), hex(k)
    assert p2(L.bpt_ec_madd, None, pts[0]) == pts[0]
    assert p2(L.bpt_ec_mul, pts[0], b32(Q)) is None
    a = pts[0]
    assert p2(L.bpt_ec_dbl_add, a, ec.neg(oracle_c.add(a, a))) is None
    assert p2(L.bpt_ec_dbl_add, a, oracle_c.add(a, a)) == oracle_c.mul(a, 4)
    o = ctypes.create_string_buffer(64)
    assert L.bpt_ec_decompress(o, b"\x02" + (5).to_bytes(32, "big")) == -1
    assert L.bpt_ec_on_curve((1).to_bytes(32, "big") * 2) == 0


def test_sha256(L):
    rng = random.Random(10)
    for n in [0, 1, 55, 56, 63, 64, 65, 130, 134, 199, 264, 1000]:
        d = bytes(rng.randrange(256) for _ in range(n))
        o = ctypes.create_string_buffer(32)
        L.bpt_sha256(o, d, n)
        assert o.raw == hashlib.sha256(d).digest(), n


def test_fq_glv_split(L):
    """fq_glv_split (fq.h): k = k1 + k2 * lambda (mod q) with the rounded quotients of libsecp256k1's scalar_split_lambda;
    magnitudes and signs against a Python model of the same formulas, on edge values and random scalars."""
    G1 = 0x3086D221A7D46BCDE86C90E49284EB153DAA8A1471E8CA7FE893209A45DBB031
    G2 = 0xE4437ED6010E88286F547FA90ABFE4C4221208AC9DF506C61571B4AE8AC47F71
    A1 = 0x3086D221A7D46BCDE86C90E49284EB15          # = b2
    MB1 = 0xE4437ED6010E88286F547FA90ABFE4C3         # = -b1
    A2 = 0x114CA50F7A8E2F3F657C1108D9D44CFD8
    LAM = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
    assert (A1 + (-MB1) * LAM) % Q == 0 and (A2 + A1 * LAM) % Q == 0      # the lattice basis (a1, b1), (a2, b2)
    rng = random.Random(11)
    edge = [0, 1, 2, Q - 1, Q - 2, (Q - 1) // 2, (Q + 1) // 2, 2**128, 2**128 - 1, 2**255, 2**255 - 19, LAM, Q - LAM, LAM - 1, 2**192 + 12345]
    for k in edge + [rng.randrange(Q) for _ in range(3000)] + [rng.randrange(2**130) for _ in range(200)]:
        c1 = (k * G1 + 2**383) >> 384
        c2 = (k * G2 + 2**383) >> 384
        k2 = c1 * MB1 - c2 * A1          # -c1 b1 - c2 b2
        k1 = k - c1 * A1 - c2 * A2
        assert (k1 + k2 * LAM - k) % Q == 0 and abs(k1) < 2**129 and abs(k2) < 2**129
        o1, o2 = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
        signs = L.bpt_fq_glv(o1, o2, b32(k))
        assert int.from_bytes(o1.raw, "big") == abs(k1) and int.from_bytes(o2.raw, "big") == abs(k2), hex(k)
        # (a zero half may carry either sign)
        assert (signs & 1) == (1 if k1 < 0 else 0) or k1 == 0
        assert ((signs >> 1) & 1) == (1 if k2 < 0 else 0) or k2 == 0
