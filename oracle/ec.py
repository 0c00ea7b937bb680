"""secp256k1 group arithmetic for the oracle (test infrastructure, see oracle/__init__.py).

Restates what curv-kzen 0.10's ``Point<Secp256k1>`` / ``Scalar<Secp256k1>`` operators compute
(SURVEY.md appendix A.2/A.3): canonical affine group elements, infinity representable.
A point is ``None`` (infinity) or an ``(x, y)`` tuple of Python ints.

Two engines with identical results:
  * pure Python ints (always available; used to validate the C engine);
  * ``oracle/_build/liboracle.so`` (plain C, 4x64-bit limbs, see oracle/csrc/secp256k1_ref.c),
    used when built because the reference-structured MSM (one full scalar multiplication per
    term, range_proof.rs:772-774) is ~100x faster in C.
"""
import ctypes
import os

P = 2**256 - 2**32 - 977
Q = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
GX = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
GY = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8
G = (GX, GY)
B = 7


def on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return 0 <= x < P and 0 <= y < P and (y * y - x * x * x - B) % P == 0


# ---------------------------------------------------------------- pure python engine
def _jac_dbl(X, Y, Z):
    if Y == 0 or Z == 0:
        return (0, 1, 0)
    S = 4 * X * Y * Y % P
    M = 3 * X * X % P
    X3 = (M * M - 2 * S) % P
    Y3 = (M * (S - X3) - 8 * Y * Y * Y * Y) % P
    Z3 = 2 * Y * Z % P
    return (X3, Y3, Z3)


def _jac_add_affine(X1, Y1, Z1, x2, y2):
    if Z1 == 0:
        return (x2, y2, 1)
    Z1Z1 = Z1 * Z1 % P
    U2 = x2 * Z1Z1 % P
    S2 = y2 * Z1 * Z1Z1 % P
    H = (U2 - X1) % P
    R = (S2 - Y1) % P
    if H == 0:
        if R == 0:
            return _jac_dbl(X1, Y1, Z1)
        return (0, 1, 0)
    HH = H * H % P
    HHH = H * HH % P
    V = X1 * HH % P
    X3 = (R * R - HHH - 2 * V) % P
    Y3 = (R * (V - X3) - Y1 * HHH) % P
    Z3 = Z1 * H % P
    return (X3, Y3, Z3)


def _to_affine(X, Y, Z):
    if Z == 0:
        return None
    zi = pow(Z, -1, P)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


def py_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    return _to_affine(*_jac_add_affine(a[0], a[1], 1, b[0], b[1]))


def py_neg(a):
    if a is None:
        return None
    return (a[0], (-a[1]) % P)


def py_mul(pt, k):
    k %= Q
    if pt is None or k == 0:
        return None
    acc = (0, 1, 0)
    for bit in bin(k)[2:]:
        acc = _jac_dbl(*acc)
        if bit == "1":
            acc = _jac_add_affine(*acc, pt[0], pt[1])
    return _to_affine(*acc)


def py_msm_naive(points, scalars, init=None):
    acc = init
    for pt, k in zip(points, scalars):
        acc = py_add(acc, py_mul(pt, k))
    return acc


def py_decompress(b33):
    """33-byte SEC1 compressed -> point, or raises ValueError (curv Point::from_bytes)."""
    if len(b33) != 33 or b33[0] not in (2, 3):
        raise ValueError("bad compressed point")
    x = int.from_bytes(b33[1:], "big")
    if x >= P:
        raise ValueError("x out of range")
    y2 = (x * x * x + B) % P
    y = pow(y2, (P + 1) // 4, P)
    if y * y % P != y2:
        raise ValueError("not on curve")
    if (y & 1) != (b33[0] & 1):
        y = P - y
    return (x, y)


# ---------------------------------------------------------------- C engine (optional)
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def _load():
    global _lib
    if _lib is not None or not os.path.exists(_LIB_PATH):
        return _lib
    lib = ctypes.CDLL(_LIB_PATH)
    u8p = ctypes.c_char_p
    lib.ref_point_mul.argtypes = [u8p, u8p, u8p]
    lib.ref_point_mul.restype = None
    lib.ref_point_add.argtypes = [u8p, u8p, u8p]
    lib.ref_point_add.restype = None
    lib.ref_msm_naive.argtypes = [u8p, u8p, u8p, ctypes.c_size_t, u8p]
    lib.ref_msm_naive.restype = None
    lib.ref_decompress.argtypes = [u8p, u8p]
    lib.ref_decompress.restype = ctypes.c_int
    lib.ref_scalar_mul_batch.argtypes = [u8p, u8p, u8p, ctypes.c_size_t]
    lib.ref_scalar_mul_batch.restype = None
    _lib = lib
    return _lib


def have_c():
    return _load() is not None


def enc64(pt):
    """Boundary encoding: x||y big-endian, infinity = 64 zero bytes."""
    if pt is None:
        return b"\0" * 64
    return pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")


def dec64(b):
    b = bytes(b)
    if b == b"\0" * 64:
        return None
    return (int.from_bytes(b[:32], "big"), int.from_bytes(b[32:], "big"))


def compress(pt):
    """curv Point::to_bytes(true): 33-byte SEC1."""
    if pt is None:
        raise ValueError("infinity has no compressed encoding")
    return bytes([2 + (pt[1] & 1)]) + pt[0].to_bytes(32, "big")


def uncompressed(pt):
    """curv Point::to_bytes(false): 0x04 || X || Y (65 bytes)."""
    if pt is None:
        raise ValueError("infinity has no uncompressed encoding")
    return b"\x04" + pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")


def mul(pt, k):
    """`&Point * &Scalar` (a1 in SURVEY.md §8a)."""
    lib = _load()
    if lib is None:
        return py_mul(pt, k)
    out = ctypes.create_string_buffer(64)
    lib.ref_point_mul(out, enc64(pt), (k % Q).to_bytes(32, "big"))
    return dec64(out.raw)


def add(a, b):
    """`Point + Point` (a2)."""
    lib = _load()
    if lib is None:
        return py_add(a, b)
    out = ctypes.create_string_buffer(64)
    lib.ref_point_add(out, enc64(a), enc64(b))
    return dec64(out.raw)


def neg(a):
    return py_neg(a)


def sub(a, b):
    """`Point - Point`."""
    return add(a, neg(b))


def msm_naive(points, scalars, init=None):
    """The reference's fold idiom `(0..n).map(|i| &p[i] * &s[i]).fold(init, |acc, x| acc + x)`
    (range_proof.rs:772-774): one full scalar multiplication + one affine addition per term."""
    lib = _load()
    if lib is None:
        return py_msm_naive(points, scalars, init)
    n = len(points)
    pb = b"".join(enc64(p) for p in points)
    sb = b"".join((k % Q).to_bytes(32, "big") for k in scalars)
    out = ctypes.create_string_buffer(64)
    lib.ref_msm_naive(out, pb, sb, n, enc64(init))
    return dec64(out.raw)


def mul_batch(points, scalars):
    """n independent k_i * P_i (hi_tag idiom, range_proof.rs:231-233)."""
    lib = _load()
    if lib is None:
        return [py_mul(p, k) for p, k in zip(points, scalars)]
    n = len(points)
    pb = b"".join(enc64(p) for p in points)
    sb = b"".join((k % Q).to_bytes(32, "big") for k in scalars)
    out = ctypes.create_string_buffer(64 * n)
    lib.ref_scalar_mul_batch(out, pb, sb, n)
    raw = out.raw
    return [dec64(raw[64 * i:64 * i + 64]) for i in range(n)]


def decompress(b33):
    return py_decompress(bytes(b33))
