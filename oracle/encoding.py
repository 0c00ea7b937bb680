"""Canonical proof byte encoding defined by this build (test infrastructure).

The reference derives serde Serialize/Deserialize (range_proof.rs:32, inner_product.rs:27,
range_proof_wip.rs:94, weighted_inner_product.rs:35) but no test or bench serializes anything,
so the byte format is ours: struct field order of the reference, points as 33-byte SEC1
compressed (curv `to_bytes(true)`), scalars/BigInts as 32-byte big-endian.

  BP  : A S T1 T2 | tau_x miu tx | L_1..L_k | R_1..R_k | a_tag b_tag            = 292 + 66k bytes
  BP+ : A | L_1..L_k | R_1..R_k | a_tag b_tag (points) | r' s' delta'           = 195 + 66k bytes
"""
from . import ec
from .inner_product import InnerProductArg
from .range_proof import RangeProof
from .range_proof_wip import RangeProofWIP
from .weighted_inner_product import WeightedInnerProdArg


def _s(v):
    return (v % ec.Q).to_bytes(32, "big")


def bp_proof_len(k):
    return 292 + 66 * k


def bpplus_proof_len(k):
    return 195 + 66 * k


def encode_bp(p):
    ipp = p.inner_product_proof
    out = [ec.compress(p.A), ec.compress(p.S), ec.compress(p.T1), ec.compress(p.T2),
           _s(p.tau_x), _s(p.miu), _s(p.tx)]
    out += [ec.compress(x) for x in ipp.L]
    out += [ec.compress(x) for x in ipp.R]
    out += [_s(ipp.a_tag), _s(ipp.b_tag)]
    return b"".join(out)


def decode_bp(b):
    b = bytes(b)
    k, rem = divmod(len(b) - 292, 66)
    if len(b) < 292 or rem:
        raise ValueError("bad BP proof length")
    pos = 0

    def pt():
        nonlocal pos
        v = ec.decompress(b[pos:pos + 33])
        pos += 33
        return v

    def sc():
        nonlocal pos
        v = int.from_bytes(b[pos:pos + 32], "big")
        pos += 32
        return v

    A, S, T1, T2 = pt(), pt(), pt(), pt()
    tau_x, miu, tx = sc(), sc(), sc()
    L = [pt() for _ in range(k)]
    R = [pt() for _ in range(k)]
    a_tag, b_tag = sc(), sc()
    return RangeProof(A, S, T1, T2, tau_x, miu, tx, InnerProductArg(L, R, a_tag, b_tag))


def encode_bpplus(p):
    w = p.weighted_inner_product_proof
    out = [ec.compress(p.A)]
    out += [ec.compress(x) for x in w.L]
    out += [ec.compress(x) for x in w.R]
    out += [ec.compress(w.a_tag), ec.compress(w.b_tag), _s(w.r_prime), _s(w.s_prime), _s(w.delta_prime)]
    return b"".join(out)


def decode_bpplus(b):
    b = bytes(b)
    k, rem = divmod(len(b) - 195, 66)
    if len(b) < 195 or rem:
        raise ValueError("bad BP+ proof length")
    pos = 0

    def pt():
        nonlocal pos
        v = ec.decompress(b[pos:pos + 33])
        pos += 33
        return v

    def sc():
        nonlocal pos
        v = int.from_bytes(b[pos:pos + 32], "big")
        pos += 32
        return v

    A = pt()
    L = [pt() for _ in range(k)]
    R = [pt() for _ in range(k)]
    a_tag, b_tag = pt(), pt()
    r_prime, s_prime, delta_prime = sc(), sc(), sc()
    return RangeProofWIP(A, WeightedInnerProdArg(L, R, a_tag, b_tag, r_prime, s_prime, delta_prime))


def encode_ipa(ipp):
    """Standalone InnerProductArg (inner_product.rs:28-33): L_1..L_k | R_1..R_k | a_tag b_tag = 66k + 64 bytes (the tail of the BP encoding)."""
    return b"".join([ec.compress(x) for x in ipp.L] + [ec.compress(x) for x in ipp.R] + [_s(ipp.a_tag), _s(ipp.b_tag)])


def encode_wipa(w):
    """Standalone WeightedInnerProdArg (weighted_inner_product.rs:36-44): L[k] | R[k] | a_tag b_tag (points) | r' s' delta' = 66k + 162 bytes."""
    return b"".join([ec.compress(x) for x in w.L] + [ec.compress(x) for x in w.R] + [ec.compress(w.a_tag), ec.compress(w.b_tag),
                                                                                      _s(w.r_prime), _s(w.s_prime), _s(w.delta_prime)])
