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
@pytest.mark.parametrize("lanes", [4, 8, 16, 64, -4, -8])
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


# ---- random-linear-combination batch verification (bp_rangeproof_verify_batch_rlc) ------------------
def _c3(ctx, window_bits=8):
    b = load("bp_c3_batch_n64_m1.json")
    gens = bpa.Gens(ctx, bytes.fromhex(b["g_vec"]), bytes.fromhex(b["h_vec"]), bytes.fromhex(b["G"]), bytes.fromhex(b["H"]), window_bits)
    return b, gens, bytearray(bytes.fromhex(b["proofs"])), bytes.fromhex(b["commitments"]), 292 + 66 * 6


@pytest.mark.parametrize("count", [1, 3, 256])
def test_rlc_all_valid_takes_the_combined_path(ctx, count):
    b, gens, proofs, coms, plen = _c3(ctx)
    try:
        for seed in (b"\x00" * 32, bytes(range(32)), b"\xff" * 32):
            got, fb = ctx.rangeproof_verify_batch_rlc(gens, 64, 1, bytes(proofs[:count * plen]), coms[:count * 64], seed)
            assert got == [0] * count and not fb
    finally:
        gens.close()


def test_rlc_with_invalid_proofs_gives_the_per_proof_verdicts(ctx):
    """A batch holding invalid proofs fails the combined check and falls back: verdict bytes equal
    bp_rangeproof_verify_batch's (the oracle's), whichever proofs were tampered and however."""
    b, gens, proofs, coms, plen = _c3(ctx)
    try:
        cnt = b["count"]
        for tampered in ([7], [0], [cnt - 1], list(range(3, cnt, 5))):
            p = bytearray(proofs)
            for i in tampered:
                p[i * plen + 4 * 33 + 64 + 31] ^= 1  # tx bit
            want = [3 if i in tampered else 0 for i in range(cnt)]
            got, fb = ctx.rangeproof_verify_batch_rlc(gens, 64, 1, bytes(p), coms, b"seed" * 8)
            assert fb and got == want
            assert got == ctx.rangeproof_verify_batch(gens, 64, 1, bytes(p), coms)
        # a single wrong commitment (valid point, wrong statement)
        c = bytearray(coms)
        c[64 * 5:64 * 6] = coms[64 * 6:64 * 7]
        got, fb = ctx.rangeproof_verify_batch_rlc(gens, 64, 1, bytes(proofs), bytes(c), b"seed" * 8)
        assert fb and got == [3 if i == 5 else 0 for i in range(cnt)]
    finally:
        gens.close()


def test_rlc_malformed_proofs_are_left_out_of_the_combination(ctx):
    """Undeserialisable proofs (bad point prefix, scalar >= q) get 255 and do not spoil the combined check."""
    b, gens, proofs, coms, plen = _c3(ctx)
    try:
        cnt = 64
        p = bytearray(proofs[:cnt * plen])
        p[9 * plen] = 7  # A of proof 9: invalid SEC1 prefix
        p[20 * plen + 4 * 33:20 * plen + 4 * 33 + 32] = b"\xff" * 32  # tau_x of proof 20 >= q
        want = [255 if i in (9, 20) else 0 for i in range(cnt)]
        got, fb = ctx.rangeproof_verify_batch_rlc(gens, 64, 1, bytes(p), coms[:cnt * 64], b"\x11" * 32)
        assert got == want and not fb
        assert ctx.rangeproof_verify_batch(gens, 64, 1, bytes(p), coms[:cnt * 64]) == want
    finally:
        gens.close()


@pytest.mark.parametrize("idx", range(5))
def test_rlc_small_cases_all_shapes(ctx, idx):
    """n, m shapes of the golden cases (m > 1: several commitments per proof), valid proof repeated + tampered ones."""
    case = bp_cases()[idx]
    n, m = case["n"], case["m"]
    gens = _gens(ctx, n, m, window_bits=8)
    try:
        good_p = bytes.fromhex(case["proof"])
        good_c = b"".join(bytes.fromhex(c) for c in case["commitments"])
        got, fb = ctx.rangeproof_verify_batch_rlc(gens, n, m, good_p * 5, good_c * 5, b"\x42" * 32)
        assert got == [0] * 5 and not fb
        proofs = [good_p] + [bytes.fromhex(t["proof"]) for t in case["tampered"]]
        coms = [good_c] + [b"".join(bytes.fromhex(c) for c in t["commitments"]) for t in case["tampered"]]
        want = [VERDICT_CODE[case["verdict"]]] + [VERDICT_CODE[t["verdict"]] for t in case["tampered"]]
        got, fb = ctx.rangeproof_verify_batch_rlc(gens, n, m, b"".join(proofs), b"".join(coms), b"\x42" * 32)
        assert got == want
    finally:
        gens.close()


# ---- Bulletproofs+ batch verification (bp_wip_rangeproof_verify_batch) ------------------------------
def _plus_gens(ctx, n, m, window_bits=8):
    g, h, G, H = bpa.generate_bases(1, bpa.KZEN_SEED, n, m)
    return bpa.Gens(ctx, g, h, G, H, window_bits), G, H


def _case_batch(case):
    proofs = [bytes.fromhex(case["proof"])] + [bytes.fromhex(t["proof"]) for t in case["tampered"]]
    coms = [b"".join(bytes.fromhex(c) for c in case["commitments"])] + [
        b"".join(bytes.fromhex(c) for c in t["commitments"]) for t in case["tampered"]]
    want = [VERDICT_CODE[case["verdict"]]] + [VERDICT_CODE[t["verdict"]] for t in case["tampered"]]
    return proofs, coms, want


@pytest.mark.parametrize("idx", range(4))
@pytest.mark.parametrize("lanes", [0, 4, 16, 64, -4, -8])
def test_bpplus_batch_cases_valid_and_tampered(ctx, idx, lanes):
    """`RangeProofWIP::aggregated_verify` (range_proof_wip.rs:388-609) on the GPU for a batch: the golden proofs
    (n*m = 8 .. 256) and their tampered variants get the oracle's verdicts, at every lanes-per-proof setting."""
    from _golden import bpplus_cases
    case = bpplus_cases()[idx]
    n, m = case["n"], case["m"]
    gens, _, _ = _plus_gens(ctx, n, m)
    ctx.set_verify_lanes(lanes)
    try:
        proofs, coms, want = _case_batch(case)
        # the same proofs several times over, so that one launch holds more than one workgroup's worth of lanes
        reps = 5
        got = ctx.wip_rangeproof_verify_batch(gens, n, m, b"".join(proofs) * reps, b"".join(coms) * reps)
        assert got == want * reps, [t["name"] for t in case["tampered"]]
    finally:
        ctx.set_verify_lanes(0)
        gens.close()


def test_bpplus_batch_c4_n64_m64(ctx):
    """BASELINE.json configs[3] shape (n=64, m=64: 8284-term MSM per proof) through the batch verifier."""
    case = load("bpplus_c4_n64_m64.json")
    gens, _, _ = _plus_gens(ctx, 64, 64, window_bits=0)
    try:
        proofs, coms, want = _case_batch(case)
        assert ctx.wip_rangeproof_verify_batch(gens, 64, 64, b"".join(proofs), b"".join(coms)) == want
    finally:
        gens.close()


def test_bpplus_batch_many_distinct_proofs(ctx, oracle_c):
    """96 distinct n=64 m=1 Bulletproofs+ proofs (made by bp_wip_rangeproof_prove, whose bytes the protocol tests pin
    to the oracle) with tampered, malformed and out-of-range ones in between: the batch verdicts equal the
    single-proof `aggregated_verify` verdicts (bp_wip_rangeproof_verify mode 2) and the oracle's on a sample."""
    import random
    from oracle import tape
    from oracle.range_proof_wip import StatementRP
    from oracle import encoding
    rng = random.Random(515)
    n, m, k = 64, 1, 6
    plen = 195 + 66 * k
    gens, G, H = _plus_gens(ctx, n, m, window_bits=10)
    Gp, Hp = ec.dec64(G), ec.dec64(H)
    try:
        proofs, coms, kinds = [], [], []
        for i in range(96):
            tb = tape.make_tape_bytes("plusbatch%d" % i, m + 1 + 2 * k + 4)
            v = rng.randrange(1 << n) if i % 11 != 7 else (1 << n) + 5  # out of range: proof does not verify
            r = int.from_bytes(tb[:32], "big")
            com = ec.enc64(oracle_c.add(oracle_c.mul(Gp, v), oracle_c.mul(Hp, r)))
            pf = bytearray(ctx.wip_rangeproof_prove(gens, n, v.to_bytes(32, "big"), tb[:32], tb[32:]))
            kind = "ok" if i % 11 != 7 else "oor"
            if i % 7 == 3:
                pf[33 * (2 * k + 3) + 40] ^= 0x10  # s' bit
                kind = "tampered"
            elif i % 13 == 5:
                pf[33 * 2 + 9] ^= 1  # L_1 x coordinate: another point or not a point at all
                kind = "point"
            elif i % 17 == 9:
                pf[33 * (2 * k + 3):33 * (2 * k + 3) + 32] = b"\xff" * 32  # r' >= q: would not deserialize
                kind = "scalar"
            proofs.append(bytes(pf))
            coms.append(com)
            kinds.append(kind)
        got = ctx.wip_rangeproof_verify_batch(gens, n, m, b"".join(proofs), b"".join(coms))
        for i in range(96):
            if kinds[i] == "ok":
                assert got[i] == 0, i
            elif kinds[i] in ("oor", "tampered"):
                assert got[i] == 3, (i, kinds[i])
            elif kinds[i] == "scalar":
                assert got[i] == 255, i
            else:  # the x coordinate of L_2 (bytes 66..99) changed: the oracle says whether it still decompresses (then 3) or not (255)
                try:
                    ec.decompress(proofs[i][66:99])
                    assert got[i] == 3, i
                except ValueError:
                    assert got[i] == 255, i
        # single-proof path on every well-formed one
        for i in range(96):
            if got[i] != 255:
                assert ctx.wip_rangeproof_verify(gens, n, m, proofs[i], coms[i], 2) == got[i], (i, kinds[i])
        # the oracle on a sample (one of each kind that deserializes)
        stmt = StatementRP.generate_bases(curv.KZEN_SEED, m, n)
        seen = set()
        for i in range(96):
            if kinds[i] in seen or got[i] == 255:
                continue
            seen.add(kinds[i])
            pr = encoding.decode_bpplus(proofs[i])
            assert VERDICT_CODE[pr.aggregated_verify(stmt, [ec.dec64(coms[i])])] == got[i], (i, kinds[i])
    finally:
        gens.close()


def test_bpplus_batch_rejects_bad_sizes(ctx):
    gens, _, _ = _plus_gens(ctx, 8, 1, window_bits=4)
    try:
        with pytest.raises(bpa.BpError) as e:
            ctx.wip_rangeproof_verify_batch(gens, 16, 1, b"\0" * (195 + 66 * 4), b"\0" * 64)  # nm != gens nm
        assert e.value.code == -1
    finally:
        gens.close()


@pytest.mark.parametrize("idx", range(4))
def test_bpplus_rlc_cases(ctx, idx):
    """bp_wip_rangeproof_verify_batch_rlc: the valid golden proof repeated passes the combined check (no fallback);
    with the tampered variants in the batch the per-proof kernel decides and the verdicts are the oracle's."""
    from _golden import bpplus_cases
    case = bpplus_cases()[idx]
    n, m = case["n"], case["m"]
    gens, _, _ = _plus_gens(ctx, n, m)
    try:
        proofs, coms, want = _case_batch(case)
        seed = bytes(range(7, 39))
        got, fb = ctx.wip_rangeproof_verify_batch_rlc(gens, n, m, proofs[0] * 6, coms[0] * 6, seed)
        assert got == [0] * 6 and not fb
        got, fb = ctx.wip_rangeproof_verify_batch_rlc(gens, n, m, b"".join(proofs) * 3, b"".join(coms) * 3, seed)
        assert got == want * 3 and fb
    finally:
        gens.close()


def test_bpplus_rlc_many_distinct_and_malformed(ctx, oracle_c):
    """80 distinct valid n=64 m=1 Bulletproofs+ proofs pass the combined check with and without malformed proofs among
    them (weight 0, verdict 255); one tampered proof sends the batch to the per-proof path with identical verdicts."""
    from oracle import tape
    n, m, k = 64, 1, 6
    gens, G, H = _plus_gens(ctx, n, m, window_bits=10)
    Gp, Hp = ec.dec64(G), ec.dec64(H)
    try:
        proofs, coms = [], []
        for i in range(80):
            tb = tape.make_tape_bytes("plusrlc%d" % i, m + 1 + 2 * k + 4)
            v = (i * 0x9E3779B97F4A7C15) & ((1 << n) - 1)
            r = int.from_bytes(tb[:32], "big")
            coms.append(ec.enc64(oracle_c.add(oracle_c.mul(Gp, v), oracle_c.mul(Hp, r))))
            proofs.append(ctx.wip_rangeproof_prove(gens, n, v.to_bytes(32, "big"), tb[:32], tb[32:]))
        seed = b"\x5a" * 32
        got, fb = ctx.wip_rangeproof_verify_batch_rlc(gens, n, m, b"".join(proofs), b"".join(coms), seed)
        assert got == [0] * 80 and not fb
        bad = list(proofs)
        bad[11] = b"\x02" + b"\x00" * 31 + b"\x05" + bad[11][33:]  # A: x = 5 is not on the curve -> would not deserialize
        bad[40] = bad[40][:33 * 15] + b"\xff" * 32 + bad[40][33 * 15 + 32:]  # r' >= q
        got, fb = ctx.wip_rangeproof_verify_batch_rlc(gens, n, m, b"".join(bad), b"".join(coms), seed)
        assert not fb and got == [255 if i in (11, 40) else 0 for i in range(80)]
        t = bytearray(bad[63])
        t[33 * 15 + 64 + 20] ^= 4  # delta'
        bad[63] = bytes(t)
        got, fb = ctx.wip_rangeproof_verify_batch_rlc(gens, n, m, b"".join(bad), b"".join(coms), seed)
        assert fb and got == [255 if i in (11, 40) else (3 if i == 63 else 0) for i in range(80)]
        assert got == ctx.wip_rangeproof_verify_batch(gens, n, m, b"".join(bad), b"".join(coms))
    finally:
        gens.close()


def _items(case):
    out = [(bytes.fromhex(case["proof"]), b"".join(bytes.fromhex(c) for c in case["commitments"]), VERDICT_CODE[case.get("verdict", "Ok")])]
    for t in case["tampered"]:
        out.append((bytes.fromhex(t["proof"]), b"".join(bytes.fromhex(c) for c in t["commitments"]), VERDICT_CODE[t["verdict"]]))
    return out


def test_bp_mixed_aggregation_sizes_in_one_call(ctx):
    """bp_rangeproof_verify_batch_mixed (proof_off[] + m_per_proof[]): n=64 proofs with m = 1, 4 and 16 over ONE
    bp_gens of the largest size (each proof uses the generator prefixes a caller of the reference would slice,
    range_proof.rs:556-557), interleaved, honest / tampered / malformed: verdicts == the oracle's fixture verdicts ==
    the per-proof path with generators of each proof's own size."""
    b = load("bp_c3_batch_n64_m1.json")
    plen1 = 292 + 66 * 6
    c3 = [(bytes.fromhex(b["proofs"])[i * plen1:(i + 1) * plen1], bytes.fromhex(b["commitments"])[i * 64:(i + 1) * 64], 0) for i in range(5)]
    sets = {1: _items(bp_cases()[3]) + c3, 4: _items(bp_cases()[4]), 16: _items(load("bp_c2_n64_m16.json"))}
    order = []
    for i in range(max(len(v) for v in sets.values())):
        for m in (4, 16, 1):
            if i < len(sets[m]):
                order.append((m,) + sets[m][i])
    gens = _gens(ctx, 64, 16, window_bits=7)
    try:
        got = ctx.rangeproof_verify_batch_mixed(gens, 64, [o[1] for o in order], [o[0] for o in order], b"".join(o[2] for o in order))
        assert got == [o[3] for o in order]
        assert {0, 3, 255} <= set(got)
        # m-homogeneous calls on a prefix of the same gens give the same bytes as gens of the exact size
        small = _gens(ctx, 64, 1, window_bits=7)
        try:
            for m, pf, cm, want in order:
                if m == 1:
                    assert ctx.rangeproof_verify(small, 64, 1, pf, cm, 2) == want
        finally:
            small.close()
        with pytest.raises(bpa.BpError):   # m = 32 exceeds the generators: the reference's assert_eq!(g_vec.len(), nm)
            ctx.rangeproof_verify_batch_mixed(gens, 64, [order[0][1]], [32], order[0][2] * 32)
        with pytest.raises(bpa.BpError):   # proof length does not match its m
            ctx.rangeproof_verify_batch_mixed(gens, 64, [sets[4][0][0]], [1], sets[4][0][1][:64])
        assert ctx.rangeproof_verify_batch_mixed(gens, 64, [], [], b"") == []
    finally:
        gens.close()


def test_mixed_batch_rejects_more_than_2pow16_bits(ctx):
    """the mixed entry points apply the k <= 16 cap of every other batch entry point (the constants kernels hold
    k + 2 prefix scalars per lane).  Generators with n*m = 2^17 (a narrow 2-bit table: 4 GB) and one n=64 m=2048 'proof' of
    the right length: BP_EINVAL, no launch."""
    import hashlib
    import bulletproofs_amd as bpa
    nm = 1 << 17
    pts = b""
    for s0 in range(0, 2 * nm + 2, 1 << 16):
        cnt = min(1 << 16, 2 * nm + 2 - s0)
        pts += ctx.fixed_base_mul_batch(b"".join(hashlib.sha256(b"r3-gens-%d" % i).digest() for i in range(s0, s0 + cnt)))
    gens = bpa.Gens(ctx, pts[:64 * nm], pts[64 * nm:128 * nm], pts[128 * nm:128 * nm + 64], pts[128 * nm + 64:], 2)
    try:
        for wip in (False, True):
            plen = (195 if wip else 292) + 66 * 17
            with pytest.raises(bpa.BpError) as e:
                ctx.rangeproof_verify_batch_mixed(gens, 64, [bytes(plen)], [2048], bytes(64 * 2048), wip=wip)
            assert e.value.code == -1
    finally:
        gens.close()


def test_bpplus_mixed_aggregation_sizes_in_one_call(ctx, oracle_c):
    """bp_wip_rangeproof_verify_batch_mixed: Bulletproofs+ n=16 proofs with m = 1, 2, 4 (GPU-made, pinned by
    test_gpu_proto.py) + the golden n=16 m=2 case with its tampered variants, one call over the m=4 generators; verdicts ==
    the per-proof path with generators of each proof's own size == the oracle's on the golden ones."""
    from oracle import tape
    from _golden import bpplus_cases
    n = 16
    items = []
    per_m_gens = {}
    for m in (1, 2, 4):
        g, h, G, H = bpa.generate_bases(1, bpa.KZEN_SEED, n, m)
        per_m_gens[m] = bpa.Gens(ctx, g, h, G, H, 6)
        k = (n * m).bit_length() - 1
        for t in range(3):
            tb = tape.make_tape_bytes("mixplus-%d-%d" % (m, t), m + 1 + 2 * k + 4)
            secs = b"".join(((t * 977 + j * 31 + 5) % (1 << n)).to_bytes(32, "big") for j in range(m))
            pf = ctx.wip_rangeproof_prove(per_m_gens[m], n, secs, tb[:32 * m], tb[32 * m:])
            cm = b"".join(ctx.msm(G + H, secs[32 * j:32 * j + 32] + tb[32 * j:32 * j + 32], None) for j in range(m))
            if t == 2:
                pf = pf[:-1] + bytes([pf[-1] ^ 1])
            items.append((m, pf, cm))
    case = bpplus_cases()[1]
    assert (case["n"], case["m"]) == (16, 2)
    gold = _items(case)
    items += [(2, pf, cm) for pf, cm, _ in gold]
    try:
        want = [ctx.wip_rangeproof_verify(per_m_gens[m], n, m, pf, cm, 2) for m, pf, cm in items]
        assert want[-len(gold):] == [v for _, _, v in gold]
        assert want[:9] == [0, 0, 3] * 3
        got = ctx.rangeproof_verify_batch_mixed(per_m_gens[4], n, [it[1] for it in items], [it[0] for it in items], b"".join(it[2] for it in items),
                                                wip=True)
        assert got == want
    finally:
        for g in per_m_gens.values():
            g.close()


@pytest.mark.parametrize("fmt", [bpa.SERDE_JSON, bpa.SERDE_BINCODE])
def test_serde_wire_format_feeds_the_verifier(ctx, fmt):
    """a proof that arrives in the serde encoding of `RangeProof` (curv-kzen 0.10's Point / Scalar / BigInt
    impls as restated in oracle/serde_curv.py - parity unpinned) is converted by bp_proof_from_serde and verified; a
    tampered one keeps its oracle verdict through the same route."""
    from oracle import encoding, serde_curv
    case = bp_cases()[0]
    n, m = case["n"], case["m"]
    gens = _gens(ctx, n, m, window_bits=8)
    try:
        wires, coms, want = [], [], []
        for rec in [case] + [t for t in case["tampered"] if VERDICT_CODE[t["verdict"]] != 255]:
            obj = encoding.decode_bp(bytes.fromhex(rec["proof"]))
            wires.append(serde_curv.to_json(obj) if fmt == bpa.SERDE_JSON else serde_curv.to_bincode(obj))
            coms.append(b"".join(bytes.fromhex(c) for c in rec["commitments"]))
            want.append(VERDICT_CODE[rec["verdict"]])
        proofs = [bpa.proof_from_serde(0, fmt, w) for w in wires]
        assert ctx.rangeproof_verify_batch(gens, n, m, b"".join(proofs), b"".join(coms)) == want
        assert want[0] == 0 and any(v != 0 for v in want)
    finally:
        gens.close()


def test_overlapped_device_calls_match_the_serial_form(ctx):
    """Successive asynchronous device calls alternate between the context's stream and a side lane with its own workspaces
    (bp_verify_set_overlap, default on), so two batches are in flight.  Seven calls of different sizes and tamper patterns
    with no sync in between - sizes going up and down so that both lanes' workspaces are re-allocated while the other lane
    is busy, and a host-pointer call (which must first order the context's stream after the side lane) in the middle - must
    give the verdict bytes of the same calls with the overlap off."""
    import torch
    dev = torch.device("cuda", 0)
    b, gens, proofs, coms, plen = _c3(ctx, 10)
    try:
        def batch(count, tamper_every, mal_at):
            pb, cb = bytearray(), bytearray()
            for j in range(count):
                i = (j * 11 + tamper_every) % b["count"]
                p = bytearray(proofs[i * plen:(i + 1) * plen])
                if j % tamper_every == 2:
                    p[4 * 33 + 64 + 31] ^= 1   # tx
                if j == mal_at:
                    p[33] = 7                  # S: not a SEC1 prefix
                pb += p
                cb += coms[i * 64:(i + 1) * 64]
            return pb, cb
        shapes = [(300, 3, 5), (40, 4, 0), (1500, 5, 1499), (7, 6, 3), (1500, 7, 2), (256, 3, 100), (900, 4, 899)]
        inputs = []
        for count, te, mal in shapes:
            pb, cb = batch(count, te, mal)
            inputs.append((count, torch.frombuffer(pb, dtype=torch.uint8).to(dev), torch.frombuffer(cb, dtype=torch.uint8).to(dev), bytes(pb), bytes(cb)))
        torch.cuda.synchronize()
        results = {}
        for overlap in (0, 1):
            ctx.set_verify_overlap(overlap)
            outs = [torch.full((c,), 77, dtype=torch.uint8, device=dev) for c, *_ in inputs]
            torch.cuda.synchronize()
            host = None
            for r, (count, d_p, d_c, pb, cb) in enumerate(inputs):
                ctx.rangeproof_verify_batch_device(gens, 64, 1, d_p.data_ptr(), plen, d_c.data_ptr(), count, outs[r].data_ptr())
                if r == 3:
                    host = ctx.rangeproof_verify_batch(gens, 64, 1, inputs[1][3], inputs[1][4])
            ctx.sync()
            results[overlap] = [o.cpu().tolist() for o in outs] + [host]
        assert results[1] == results[0]
        for (count, te, mal), got in zip(shapes, results[1]):
            assert got == [255 if j == mal else (3 if j % te == 2 else 0) for j in range(count)]
        assert results[1][-1] == results[1][1]
    finally:
        ctx.set_verify_overlap(1)
        gens.close()


def test_overlapped_bpplus_and_bp_calls_interleaved(ctx):
    """Bulletproofs+ and Bulletproofs device calls alternating on one context with no sync in between (so consecutive calls
    of DIFFERENT protocols, generator tables and shapes sit on the two lanes at once): the golden verdicts, overlap on
    and off."""
    import torch
    from _golden import bpplus_cases
    dev = torch.device("cuda", 0)
    plus = bpplus_cases()[1]
    bp = bp_cases()[2]
    gens_p, _, _ = _plus_gens(ctx, plus["n"], plus["m"])
    gens_b = _gens(ctx, bp["n"], bp["m"], 8)
    try:
        pp, pc, pw = _case_batch(plus)
        bp_proofs, bp_coms, bw = _case_batch(bp)
        reps = 7
        jobs = []
        for r in range(6):
            for (fn, gens, case, proofs, coms, want) in ((ctx.wip_rangeproof_verify_batch_device, gens_p, plus, pp, pc, pw),
                                                         (ctx.rangeproof_verify_batch_device, gens_b, bp, bp_proofs, bp_coms, bw)):
                k = reps + r
                d_p = torch.frombuffer(bytearray(b"".join(proofs) * k), dtype=torch.uint8).to(dev)
                d_c = torch.frombuffer(bytearray(b"".join(coms) * k), dtype=torch.uint8).to(dev)
                jobs.append((fn, gens, case["n"], case["m"], d_p, len(proofs[0]), d_c, len(proofs) * k, want * k))
        torch.cuda.synchronize()
        for overlap in (1, 0):
            ctx.set_verify_overlap(overlap)
            outs = [torch.full((j[7],), 77, dtype=torch.uint8, device=dev) for j in jobs]
            torch.cuda.synchronize()
            for (fn, gens, n, m, d_p, plen, d_c, cnt, want), o in zip(jobs, outs):
                fn(gens, n, m, d_p.data_ptr(), plen, d_c.data_ptr(), cnt, o.data_ptr())
            ctx.sync()
            for (fn, gens, n, m, d_p, plen, d_c, cnt, want), o in zip(jobs, outs):
                assert o.cpu().tolist() == want
    finally:
        ctx.set_verify_overlap(1)
        gens_p.close()
        gens_b.close()
