"""serde encodings of the reference's proof structs (test infrastructure).

`#[derive(Serialize, Deserialize)]` on RangeProof (range_proof.rs:32-42), InnerProductArg (inner_product.rs:27-33),
RangeProofWIP (range_proof_wip.rs:93-97) and WeightedInnerProdArg (weighted_inner_product.rs:35-44) makes the wire
format whatever curv-kzen 0.10 defines for Point / Scalar / BigInt.  curv is not under /root/reference and nothing in
the reference serialises a proof, so the three leaf encodings below restate curv's published serde impls
[curv-recollection, SURVEY appendix A.2] - PARITY UNPINNED:

  Point<E>   -> struct Point  { curve: &str = "secp256k1", point:  bytes = to_bytes(true) }   (33-byte SEC1)
  Scalar<E>  -> struct Scalar { curve: &str = "secp256k1", scalar: bytes = 32-byte big-endian }
  BigInt     -> human-readable serializer: hex string of to_bytes() (minimal big-endian, zero = one 00 byte);
                otherwise serialize_bytes(to_bytes())

Containers follow serde's data model: serde_json writes structs as objects in declaration order and byte strings as
arrays of numbers; bincode 1.x (default options) writes fields in order with u64 little-endian lengths before str,
bytes and Vec.  Works on the oracle's proof objects, independently of the library's codec (csrc/serde_codec.cpp).
"""
import json
import struct

from . import ec
from .inner_product import InnerProductArg
from .range_proof import RangeProof
from .range_proof_wip import RangeProofWIP
from .weighted_inner_product import WeightedInnerProdArg

CURVE = "secp256k1"


def _bigint_bytes(v):
    v = int(v)
    return v.to_bytes(max(1, (v.bit_length() + 7) // 8), "big")


# ---- serde_json -------------------------------------------------------------------------------------------------------
def _jp(pt):
    return {"curve": CURVE, "point": list(ec.compress(pt))}


def _js(v):
    return {"curve": CURVE, "scalar": list((v % ec.Q).to_bytes(32, "big"))}


def _jb(v):
    return _bigint_bytes(v).hex()


def to_json(p):
    if isinstance(p, RangeProof):
        ipp = p.inner_product_proof
        d = {"A": _jp(p.A), "S": _jp(p.S), "T1": _jp(p.T1), "T2": _jp(p.T2), "tau_x": _js(p.tau_x), "miu": _js(p.miu), "tx": _js(p.tx),
             "inner_product_proof": {"L": [_jp(x) for x in ipp.L], "R": [_jp(x) for x in ipp.R], "a_tag": _jb(ipp.a_tag), "b_tag": _jb(ipp.b_tag)}}
    else:
        w = p.weighted_inner_product_proof
        d = {"A": _jp(p.A),
             "weighted_inner_product_proof": {"L": [_jp(x) for x in w.L], "R": [_jp(x) for x in w.R], "a_tag": _jp(w.a_tag), "b_tag": _jp(w.b_tag),
                                              "r_prime": _jb(w.r_prime), "s_prime": _jb(w.s_prime), "delta_prime": _jb(w.delta_prime)}}
    return json.dumps(d, separators=(",", ":")).encode()  # serde_json::to_string: no whitespace, declaration order


def _pj(d):
    if d["curve"] != CURVE or len(d["point"]) != 33:
        raise ValueError("bad point")
    return ec.decompress(bytes(d["point"]))


def _sj(d):
    if d["curve"] != CURVE or len(d["scalar"]) != 32:
        raise ValueError("bad scalar")
    return int.from_bytes(bytes(d["scalar"]), "big")


def from_json(kind, text):
    d = json.loads(text)
    if kind == 0:
        i = d["inner_product_proof"]
        return RangeProof(_pj(d["A"]), _pj(d["S"]), _pj(d["T1"]), _pj(d["T2"]), _sj(d["tau_x"]), _sj(d["miu"]), _sj(d["tx"]),
                          InnerProductArg([_pj(x) for x in i["L"]], [_pj(x) for x in i["R"]], int(i["a_tag"], 16), int(i["b_tag"], 16)))
    w = d["weighted_inner_product_proof"]
    return RangeProofWIP(_pj(d["A"]), WeightedInnerProdArg([_pj(x) for x in w["L"]], [_pj(x) for x in w["R"]], _pj(w["a_tag"]), _pj(w["b_tag"]),
                                                           int(w["r_prime"], 16), int(w["s_prime"], 16), int(w["delta_prime"], 16)))


# ---- bincode 1.x, default options -------------------------------------------------------------------------------------
def _blob(b):
    return struct.pack("<Q", len(b)) + bytes(b)


def _bp(pt):
    return _blob(CURVE.encode()) + _blob(ec.compress(pt))


def _bs(v):
    return _blob(CURVE.encode()) + _blob((v % ec.Q).to_bytes(32, "big"))


def _bb(v):
    return _blob(_bigint_bytes(v))


def _bvec(pts):
    return struct.pack("<Q", len(pts)) + b"".join(_bp(x) for x in pts)


def to_bincode(p):
    if isinstance(p, RangeProof):
        i = p.inner_product_proof
        return b"".join([_bp(p.A), _bp(p.S), _bp(p.T1), _bp(p.T2), _bs(p.tau_x), _bs(p.miu), _bs(p.tx), _bvec(i.L), _bvec(i.R), _bb(i.a_tag), _bb(i.b_tag)])
    w = p.weighted_inner_product_proof
    return b"".join([_bp(p.A), _bvec(w.L), _bvec(w.R), _bp(w.a_tag), _bp(w.b_tag), _bb(w.r_prime), _bb(w.s_prime), _bb(w.delta_prime)])
