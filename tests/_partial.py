"""Reader / writer of the byte format of `bp_msm_partial*` (include/bpmsm.h): W = (129 + c) // c window sums, each an XYZZ
point in the library's compute form - X, Y, ZZ, ZZZ as 9 little-endian 32-bit words holding 29-bit limbs (fp.h), 144 bytes,
identity = ZZ = 0 (ec.h) - so bp_msm_partial_bytes(c) = W * 144.  Test infrastructure: the gloo test moves buffers of
exactly this format between processes; the GPU tests tie the format to the product (decode real partials, feed encoded
ones to bp_msm_combine)."""
from oracle import ec

FP_BYTES = 36
XYZZ_BYTES = 4 * FP_BYTES


def windows(c):
    return (129 + c) // c


def _fp_enc(v):
    return b"".join(((v >> (29 * i)) & 0x1FFFFFFF).to_bytes(4, "little") for i in range(9))


def _fp_dec(b):
    return sum(int.from_bytes(b[4 * i:4 * i + 4], "little") << (29 * i) for i in range(9)) % ec.P


def encode_partial(window_sums, c):
    """window_sums: W affine points (None = identity) -> bp_msm_partial_bytes(c) bytes (ZZ = ZZZ = 1)."""
    assert len(window_sums) == windows(c)
    out = bytearray()
    for p in window_sums:
        if p is None:
            out += _fp_enc(0) + _fp_enc(1) + _fp_enc(0) + _fp_enc(0)
        else:
            out += _fp_enc(p[0]) + _fp_enc(p[1]) + _fp_enc(1) + _fp_enc(1)
    return bytes(out)


def decode_partial(b, c):
    """-> W affine points (None = identity): x = X / ZZ, y = Y / ZZZ."""
    W = windows(c)
    assert len(b) == W * XYZZ_BYTES
    out = []
    for w in range(W):
        q = b[w * XYZZ_BYTES:(w + 1) * XYZZ_BYTES]
        X, Y, ZZ, ZZZ = (_fp_dec(q[i * FP_BYTES:(i + 1) * FP_BYTES]) for i in range(4))
        if ZZ == 0:
            out.append(None)
        else:
            out.append((X * pow(ZZ, -1, ec.P) % ec.P, Y * pow(ZZZ, -1, ec.P) % ec.P))
    return out


def combine(partials, nparts, c):
    """What bp_msm_combine computes: per-window sum over the parts, then sum_w 2^(c w) S_w (Horner from the top)."""
    W = windows(c)
    size = W * XYZZ_BYTES
    sums = [None] * W
    for g in range(nparts):
        for w, p in enumerate(decode_partial(partials[g * size:(g + 1) * size], c)):
            sums[w] = ec.add(sums[w], p)
    acc = None
    for w in reversed(range(W)):
        if acc is not None:
            acc = ec.mul(acc, 1 << c)
        acc = ec.add(acc, sums[w])
    return acc


def unsigned_window_sums(points, scalars, c):
    """Stand-in producer for CPU tests: S_w = sum_i digit_w(k_i) P_i with unsigned c-bit digits (scalars < 2^(c W))."""
    W = windows(c)
    assert all(0 <= k < (1 << (c * W)) for k in scalars)
    out = []
    for w in range(W):
        digs = [(k >> (c * w)) & ((1 << c) - 1) for k in scalars]
        out.append(ec.msm_naive(points, digs, None))
    return out
