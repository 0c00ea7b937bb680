"""Randomness tape (test infrastructure).

The reference draws every random scalar with `Scalar::<Secp256k1>::random()` (OS RNG, not seedable:
range_proof.rs:57-58,109-114,164-165; range_proof_wip.rs:150; weighted_inner_product.rs:106,134,
213-219).  "Same proof bytes for a fixed seed" therefore needs the draws supplied from outside in
the reference's draw order.  The tape is a flat byte string of 32-byte big-endian scalars in
[1, q); `bp_rangeproof_prove` takes the same bytes (include/bpmsm.h).

Tape generation: SHA-256 counter mode over a seed string, rejection-sampled.
"""
import hashlib

from .ec import Q


class Tape:
    def __init__(self, data):
        self.data = bytes(data)
        self.pos = 0

    def random_scalar(self):
        """Stands in for `Scalar::<Secp256k1>::random()`."""
        if self.pos + 32 > len(self.data):
            raise ValueError("randomness tape exhausted")
        v = int.from_bytes(self.data[self.pos:self.pos + 32], "big")
        self.pos += 32
        if not (0 < v < Q):
            raise ValueError("tape scalar out of range")
        return v


def make_tape_bytes(seed, count):
    out = []
    ctr = 0
    sb = seed.encode() if isinstance(seed, str) else bytes(seed)
    while len(out) < count:
        d = hashlib.sha256(sb + ctr.to_bytes(8, "big")).digest()
        ctr += 1
        v = int.from_bytes(d, "big")
        if 0 < v < Q:
            out.append(d)
    return b"".join(out)


def bp_tape_len(n, m):
    """alpha, rho, SR[nm], SL[nm], tau1, tau2 (range_proof.rs:57-58,109-114,164-165)."""
    return 2 + 2 * n * m + 2


def bpplus_tape_len(n, m):
    """alpha, then (d_L, d_R) per round, then r, s, delta, eta."""
    nm = n * m
    k = nm.bit_length() - 1
    return 1 + 2 * k + 4
