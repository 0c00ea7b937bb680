"""GPU parity: bp_rangeproof_verify_batch (device-side aggregated_verify) against the golden
fixtures and the oracle's verdicts, through the C ABI."""
import pytest

import bulletproofs_amd as bpa
from oracle import curv, ec

from _golden import VERDICT_CODE, bp_cases, load

pytestmark = pytest.mark.gpu


def _gens(ctx, n, m, window_bits=0):
    g_vec, h_vec, G, H = curv.bp_generators(n, m)
    return bpa.Gens(ctx, b"".join(ec.enc64(p) for p in g_vec), b"".join(ec.enc64(p) for p in h_vec), ec.enc64(G), ec.enc64(H), window_bits)


@pytest.mark.parametrize("idx", range(5))
@pytest.mark.parametrize("lanes", [4, 8, 16, 64])
def test_bp_cases_valid_and_tampered(ctx, idx, lanes):
    case = bp_cases()[idx]
    n, m = case["n"], case["m"]
    gens = _gens(ctx, n, m, window_bits=8)
    ctx.set_verify_lanes(lanes)
    try:
        proofs = [bytes.fromhex(case["proof"])] + [bytes.fromhex(t["proof"]) for t in case["tampered"]]
        coms = [b"".join(bytes.fromhex(c) for c in case["commitments"])] + [
            b"".join(bytes.fromhex(c) for c in t["commitments"]) for t in case["tampered"]]
        want = [VERDICT_CODE[case["verdict"]]] + [VERDICT_CODE[t["verdict"]] for t in case["tampered"]]
        got = ctx.rangeproof_verify_batch(gens, n, m, b"".join(proofs), b"".join(coms))
        assert got == want, [(t["name"]) for t in case["tampered"]]
    finally:
        ctx.set_verify_lanes(0)
        gens.close()


def test_bp_c2_n64_m16(ctx):
    case = load("bp_c2_n64_m16.json")
    gens = _gens(ctx, 64, 16)
    try:
        proofs = [bytes.fromhex(case["proof"])] + [bytes.fromhex(t["proof"]) for t in case["tampered"]]
        coms = [b"".join(bytes.fromhex(c) for c in case["commitments"])] + [
            b"".join(bytes.fromhex(c) for c in t["commitments"]) for t in case["tampered"]]
        want = [0] + [VERDICT_CODE[t["verdict"]] for t in case["tampered"]]
        assert ctx.rangeproof_verify_batch(gens, 64, 16, b"".join(proofs), b"".join(coms)) == want
    finally:
        gens.close()


@pytest.mark.parametrize("window_bits", [4, 8, 10, 16, 0])
def test_bp_c3_batch(ctx, window_bits):
    """256 distinct n=64 m=1 proofs sharing generators, with every 5th one tampered."""
    b = load("bp_c3_batch_n64_m1.json")
    gens = bpa.Gens(ctx, bytes.fromhex(b["g_vec"]), bytes.fromhex(b["h_vec"]), bytes.fromhex(b["G"]), bytes.fromhex(b["H"]), window_bits)
    try:
        proofs = bytearray(bytes.fromhex(b["proofs"]))
        coms = bytes.fromhex(b["commitments"])
        plen = 292 + 66 * 6
        cnt = b["count"]
        want = []
        for i in range(cnt):
            if i % 5 == 3:
                proofs[i * plen + 4 * 33 + 64 + 31] ^= 1  # tx bit
                want.append(3)
            else:
                want.append(0)
        got = ctx.rangeproof_verify_batch(gens, 64, 1, bytes(proofs), coms)
        assert got == want
    finally:
        gens.close()


def test_verify_rejects_bad_sizes(ctx):
    b = load("bp_c3_batch_n64_m1.json")
    gens = bpa.Gens(ctx, bytes.fromhex(b["g_vec"]), bytes.fromhex(b["h_vec"]), bytes.fromhex(b["G"]), bytes.fromhex(b["H"]), 4)
    try:
        with pytest.raises(bpa.BpError) as e:
            ctx.rangeproof_verify_batch(gens, 32, 1, b"\0" * (292 + 66 * 5), b"\0" * 64)  # nm != gens nm
        assert e.value.code == -1
    finally:
        gens.close()
