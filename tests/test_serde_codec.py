"""bp_proof_to_serde / bp_proof_from_serde against oracle/serde_curv.py on every golden proof.
Host-only entry points: they run without a GPU.  The leaf encodings restate curv-kzen 0.10's serde impls (curv is not
vendored, the reference never serialises a proof): parity with the real crate is UNPINNED; what is pinned here is
library == oracle restatement, byte for byte, both directions, both formats."""
import json

import pytest

import bulletproofs_amd as bpa
from oracle import encoding, serde_curv
import _golden

JSON, BINCODE = bpa.SERDE_JSON, bpa.SERDE_BINCODE


def _proofs():
    out = []
    for c in _golden.bp_cases():
        out.append((0, bytes.fromhex(c["proof"])))
        for t in c["tampered"]:
            if "proof" in t:
                out.append((0, bytes.fromhex(t["proof"])))
    for c in _golden.bpplus_cases():
        out.append((1, bytes.fromhex(c["proof"])))
    out.append((0, bytes.fromhex(_golden.load("bp_c2_n64_m16.json")["proof"])))
    out.append((1, bytes.fromhex(_golden.load("bpplus_c4_n64_m64.json")["proof"])))
    return out


def _decodable(kind, b):
    try:
        return encoding.decode_bp(b) if kind == 0 else encoding.decode_bpplus(b)
    except Exception:
        return None


def test_library_matches_oracle_restatement_both_formats():
    n = 0
    for kind, b in _proofs():
        obj = _decodable(kind, b)
        if obj is None:  # tampered fixture with an off-curve point: the oracle object cannot be built
            continue
        if (encoding.encode_bp(obj) if kind == 0 else encoding.encode_bpplus(obj)) != b:
            continue  # tampered fixture with a scalar >= q: no curv Scalar holds it, the oracle object is the reduced value
        js = bpa.proof_to_serde(kind, JSON, b)
        assert js == serde_curv.to_json(obj)
        bc = bpa.proof_to_serde(kind, BINCODE, b)
        assert bc == serde_curv.to_bincode(obj)
        assert bpa.proof_from_serde(kind, JSON, js) == b
        assert bpa.proof_from_serde(kind, BINCODE, bc) == b
        back = serde_curv.from_json(kind, js)
        assert (encoding.encode_bp(back) if kind == 0 else encoding.encode_bpplus(back)) == b
        n += 1
    assert n >= 12


def test_json_reader_accepts_any_field_order_and_whitespace():
    kind, b = 0, bytes.fromhex(_golden.bp_cases()[0]["proof"])
    d = json.loads(bpa.proof_to_serde(kind, JSON, b))
    shuffled = {k: d[k] for k in reversed(list(d))}
    shuffled["inner_product_proof"] = {k: d["inner_product_proof"][k] for k in reversed(list(d["inner_product_proof"]))}
    text = json.dumps(shuffled, indent=2).encode()
    assert bpa.proof_from_serde(kind, JSON, text) == b
    # BigInt hex: upper case and an odd number of digits parse to the same value
    d["inner_product_proof"]["a_tag"] = d["inner_product_proof"]["a_tag"].upper().lstrip("0") or "0"
    assert bpa.proof_from_serde(kind, JSON, json.dumps(d).encode()) == b


def test_small_bigint_is_minimal_and_zero_is_one_byte():
    c = _golden.bp_cases()[0]
    b = bytearray(bytes.fromhex(c["proof"]))
    b[-64:-32] = (5).to_bytes(32, "big")   # a_tag = 5
    b[-32:] = bytes(32)                     # b_tag = 0
    d = json.loads(bpa.proof_to_serde(0, JSON, bytes(b)))
    assert d["inner_product_proof"]["a_tag"] == "05" and d["inner_product_proof"]["b_tag"] == "00"
    bc = bpa.proof_to_serde(0, BINCODE, bytes(b))
    assert bc.endswith((1).to_bytes(8, "little") + b"\x05" + (1).to_bytes(8, "little") + b"\x00")
    assert bpa.proof_from_serde(0, JSON, json.dumps(d).encode()) == bytes(b)
    assert bpa.proof_from_serde(0, BINCODE, bc) == bytes(b)


@pytest.mark.parametrize("mutate", ["curve", "short_point", "prefix", "missing", "trailing", "bigint_long", "not_json", "lr_len"])
def test_malformed_inputs_are_rejected(mutate):
    b = bytes.fromhex(_golden.bp_cases()[0]["proof"])
    d = json.loads(bpa.proof_to_serde(0, JSON, b))
    text = None
    if mutate == "curve":
        d["S"]["curve"] = "ed25519"
    elif mutate == "short_point":
        d["T1"]["point"] = d["T1"]["point"][:32]
    elif mutate == "prefix":
        d["A"]["point"][0] = 4
    elif mutate == "missing":
        del d["tx"]
    elif mutate == "bigint_long":
        d["inner_product_proof"]["b_tag"] = "01" + "00" * 32
    elif mutate == "lr_len":
        d["inner_product_proof"]["R"] = d["inner_product_proof"]["R"][:-1]
    elif mutate == "trailing":
        text = json.dumps(d).encode() + b" x"
    elif mutate == "not_json":
        text = b"{\"A\": [1, 2"
    with pytest.raises(bpa.BpError) as e:
        bpa.proof_from_serde(0, JSON, text if text is not None else json.dumps(d).encode())
    assert e.value.code == bpa.BP_EINVAL


def test_bincode_and_canonical_rejections():
    b = bytes.fromhex(_golden.bpplus_cases()[0]["proof"])
    bc = bpa.proof_to_serde(1, BINCODE, b)
    for bad in (bc[:-1], bc + b"\x00", bc[:8] + b"SECP256K1" + bc[17:], b""):
        with pytest.raises(bpa.BpError):
            bpa.proof_from_serde(1, BINCODE, bad)
    with pytest.raises(bpa.BpError):
        bpa.proof_to_serde(1, JSON, b[:-1])           # not 195 + 66 k bytes
    with pytest.raises(bpa.BpError):
        bpa.proof_to_serde(0, JSON, b)                # wrong kind for this length
    with pytest.raises(bpa.BpError):
        bpa.proof_to_serde(1, JSON, b"\x05" + b[1:])  # no SEC1 prefix
    with pytest.raises(bpa.BpError):
        bpa.proof_to_serde(1, 7, b)                   # unknown format
