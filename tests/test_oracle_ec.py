"""Pins the oracle's group arithmetic (both engines) against OpenSSL libcrypto and the standard
secp256k1 constants (SURVEY.md §8c 'independent arithmetic oracle')."""
import ctypes
import ctypes.util
import random

import pytest

from oracle import ec


def _openssl():
    name = ctypes.util.find_library("crypto")
    if not name:
        pytest.skip("libcrypto not available")
    L = ctypes.CDLL(name)
    vp = ctypes.c_void_p
    for f, args in [("EC_GROUP_new_by_curve_name", [ctypes.c_int]), ("EC_POINT_new", [vp]), ("BN_new", []), ("BN_CTX_new", []),
                    ("BN_bin2bn", [ctypes.c_char_p, ctypes.c_int, vp])]:
        getattr(L, f).restype = vp
        getattr(L, f).argtypes = args
    L.EC_POINT_mul.argtypes = [vp, vp, vp, vp, vp, vp]
    L.EC_POINT_point2oct.argtypes = [vp, vp, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, vp]
    L.EC_POINT_point2oct.restype = ctypes.c_size_t
    L.EC_POINT_oct2point.argtypes = [vp, vp, ctypes.c_char_p, ctypes.c_size_t, vp]
    L.EC_POINT_add.argtypes = [vp, vp, vp, vp, vp]
    return L


class _Ossl:
    def __init__(self):
        self.L = _openssl()
        self.grp = self.L.EC_GROUP_new_by_curve_name(714)  # NID_secp256k1
        assert self.grp
        self.ctx = self.L.BN_CTX_new()

    def _pt(self, p):
        q = self.L.EC_POINT_new(self.grp)
        assert self.L.EC_POINT_oct2point(self.grp, q, ec.uncompressed(p), 65, self.ctx) == 1
        return q

    def _out(self, q):
        buf = ctypes.create_string_buffer(65)
        n = self.L.EC_POINT_point2oct(self.grp, q, 4, buf, 65, self.ctx)
        if n == 1:
            return None
        return (int.from_bytes(buf.raw[1:33], "big"), int.from_bytes(buf.raw[33:], "big"))

    def mul(self, p, k):
        kb = k.to_bytes(32, "big")
        bn = self.L.BN_bin2bn(kb, 32, None)
        r = self.L.EC_POINT_new(self.grp)
        assert self.L.EC_POINT_mul(self.grp, r, None, self._pt(p), bn, self.ctx) == 1
        return self._out(r)

    def add(self, a, b):
        r = self.L.EC_POINT_new(self.grp)
        assert self.L.EC_POINT_add(self.grp, r, self._pt(a), self._pt(b), self.ctx) == 1
        return self._out(r)


def test_constants():
    assert ec.on_curve(ec.G)
    assert ec.P == 2**256 - 2**32 - 977
    assert ec.py_mul(ec.G, ec.Q) is None
    assert ec.py_mul(ec.G, 2)[0] == 0xC6047F9441ED7D6D3045406E95C07CD85C778E4B8CEF3CA7ABAC09B95C709EE5


def test_against_openssl(oracle_c):
    o = _Ossl()
    rng = random.Random(11)
    pts = [ec.G]
    for _ in range(12):
        k = rng.randrange(1, ec.Q)
        p = rng.choice(pts)
        want = o.mul(p, k)
        assert ec.py_mul(p, k) == want
        assert oracle_c.mul(p, k) == want
        pts.append(want)
    for _ in range(12):
        a, b = rng.choice(pts), rng.choice(pts)
        want = o.add(a, b)
        assert ec.py_add(a, b) == want == oracle_c.add(a, b)
    a = pts[3]
    assert oracle_c.add(a, ec.neg(a)) is None and ec.py_add(a, ec.neg(a)) is None
    assert oracle_c.add(a, a) == o.mul(a, 2)


def test_engines_agree_on_msm(oracle_c):
    rng = random.Random(3)
    pts = [ec.py_mul(ec.G, rng.randrange(1, ec.Q)) for _ in range(9)] + [None]
    ks = [rng.randrange(ec.Q) for _ in range(8)] + [0, 5]
    assert oracle_c.msm_naive(pts, ks, pts[0]) == ec.py_msm_naive(pts, ks, pts[0])
    assert oracle_c.mul_batch(pts[:4], ks[:4]) == [ec.py_mul(p, k) for p, k in zip(pts[:4], ks[:4])]


def test_decompress(oracle_c):
    rng = random.Random(4)
    for _ in range(10):
        p = oracle_c.mul(ec.G, rng.randrange(1, ec.Q))
        assert ec.decompress(ec.compress(p)) == p
    with pytest.raises(ValueError):
        ec.decompress(b"\x02" + (5).to_bytes(32, "big"))  # x = 5 is not on secp256k1
