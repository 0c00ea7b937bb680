"""GPU parity of the protocol layer (C ABI: bp_rangeproof_prove / _verify, bp_wip_*, bp_fold_bases,
bp_scalar_mul_batch, bp_fixed_base_mul_batch): proof BYTES identical to the golden fixtures for the
same randomness tape, and verdicts identical to the oracle's on valid and tampered proofs."""
import random

import pytest

import bulletproofs_amd as bpa
from oracle import ec

from _golden import VERDICT_CODE, bp_cases, bpplus_cases, load

pytestmark = pytest.mark.gpu


def _gens(ctx, kind, n, m, wb=6):
    g, h, G, H = bpa.generate_bases(kind, bpa.KZEN_SEED, n, m)
    return bpa.Gens(ctx, g, h, G, H, wb)


def _be(xs):
    return b"".join(int(x, 16).to_bytes(32, "big") for x in xs)


@pytest.mark.parametrize("idx", range(5))
def test_bp_prove_bytes_match_golden(ctx, idx):
    case = bp_cases()[idx]
    n, m = case["n"], case["m"]
    gens = _gens(ctx, 0, n, m)
    try:
        proof = ctx.rangeproof_prove(gens, n, _be(case["secrets"]), _be(case["blindings"]), bytes.fromhex(case["tape"]))
        assert proof.hex() == case["proof"]
    finally:
        gens.close()


@pytest.mark.parametrize("idx", (0, 2))
def test_reference_structured_rounds_give_the_same_bytes(ctx, idx):
    """proto path 1 (hi_tag materialised, bases folded every round as inner_product.rs:140-155 does) and the
    default fixed-table path produce the same proof bytes."""
    ctx.set_proto_path(1)
    try:
        case = bp_cases()[idx]
        gens = _gens(ctx, 0, case["n"], case["m"])
        try:
            proof = ctx.rangeproof_prove(gens, case["n"], _be(case["secrets"]), _be(case["blindings"]), bytes.fromhex(case["tape"]))
            assert proof.hex() == case["proof"]
        finally:
            gens.close()
        case = bpplus_cases()[idx]
        gens = _gens(ctx, 1, case["n"], case["m"])
        try:
            proof = ctx.wip_rangeproof_prove(gens, case["n"], _be(case["secrets"]), _be(case["blindings"]), bytes.fromhex(case["tape"]))
            assert proof.hex() == case["proof"]
        finally:
            gens.close()
    finally:
        ctx.set_proto_path(0)


def test_bp_prove_c2_n64_m16(ctx):
    case = load("bp_c2_n64_m16.json")
    gens = _gens(ctx, 0, 64, 16)
    try:
        proof = ctx.rangeproof_prove(gens, 64, _be(case["secrets"]), _be(case["blindings"]), bytes.fromhex(case["tape"]))
        assert proof.hex() == case["proof"]
        coms = b"".join(bytes.fromhex(c) for c in case["commitments"])
        assert ctx.rangeproof_verify(gens, 64, 16, proof, coms, 2) == 0
        assert ctx.rangeproof_verify(gens, 64, 16, proof, coms, 1) == 0
    finally:
        gens.close()


@pytest.mark.parametrize("idx", range(4))
def test_bp_verify_modes_agree_with_oracle(ctx, idx):
    """verify == fast_verify == aggregated_verify (the reference's three verifiers) on honest and tampered proofs."""
    case = bp_cases()[idx]
    n, m = case["n"], case["m"]
    gens = _gens(ctx, 0, n, m)
    try:
        items = [(case["proof"], case["commitments"], case["verdict"])] + [(t["proof"], t["commitments"], t["verdict"]) for t in case["tampered"]]
        for pf, cm, want in items:
            coms = b"".join(bytes.fromhex(c) for c in cm)
            for mode in (0, 1, 2):
                assert ctx.rangeproof_verify(gens, n, m, bytes.fromhex(pf), coms, mode) == VERDICT_CODE[want], (mode, want)
    finally:
        gens.close()


@pytest.mark.parametrize("idx", (0, 1))
def test_recursive_verify_with_folded_bases(ctx, idx):
    """proto path 1: `verify` folds the bases every round like inner_product.rs:171-244; same verdicts."""
    case = bp_cases()[idx]
    n, m = case["n"], case["m"]
    gens = _gens(ctx, 0, n, m)
    ctx.set_proto_path(1)
    try:
        items = [(case["proof"], case["commitments"], case["verdict"])] + [(t["proof"], t["commitments"], t["verdict"]) for t in case["tampered"]]
        for pf, cm, want in items:
            coms = b"".join(bytes.fromhex(c) for c in cm)
            assert ctx.rangeproof_verify(gens, n, m, bytes.fromhex(pf), coms, 0) == VERDICT_CODE[want], want
    finally:
        ctx.set_proto_path(0)
        gens.close()


@pytest.mark.parametrize("idx", range(4))
def test_bpplus_prove_and_verify(ctx, idx):
    case = bpplus_cases()[idx]
    n, m = case["n"], case["m"]
    gens = _gens(ctx, 1, n, m)
    try:
        proof = ctx.wip_rangeproof_prove(gens, n, _be(case["secrets"]), _be(case["blindings"]), bytes.fromhex(case["tape"]))
        assert proof.hex() == case["proof"]
        items = [(case["proof"], case["commitments"], case["verdict"])] + [(t["proof"], t["commitments"], t["verdict"]) for t in case["tampered"]]
        for pf, cm, want in items:
            coms = b"".join(bytes.fromhex(c) for c in cm)
            for mode in (0, 2):
                assert ctx.wip_rangeproof_verify(gens, n, m, bytes.fromhex(pf), coms, mode) == VERDICT_CODE[want], (mode, want)
    finally:
        gens.close()


def test_out_of_range_secret_is_rejected(ctx, oracle_c):
    """range_proof.rs:955-1006 / range_proof_wip.rs:703-736 (#[should_panic]): v >= 2^n does not verify."""
    from oracle import tape
    n, m = 8, 2
    for kind, prove, verify in ((0, ctx.rangeproof_prove, ctx.rangeproof_verify), (1, ctx.wip_rangeproof_prove, ctx.wip_rangeproof_verify)):
        g, h, G, H = bpa.generate_bases(kind, bpa.KZEN_SEED, n, m)
        gens = bpa.Gens(ctx, g, h, G, H, 6)
        try:
            tb = tape.make_tape_bytes("oor%d" % kind, m + 2 * n * m + 4)
            v = [3, 1 << (n + 1)]
            r = [int.from_bytes(tb[32 * j:32 * j + 32], "big") for j in range(m)]
            Gp, Hp = ec.dec64(G), ec.dec64(H)
            coms = b"".join(ec.enc64(oracle_c.add(oracle_c.mul(Gp, v[j]), oracle_c.mul(Hp, r[j]))) for j in range(m))
            proof = prove(gens, n, b"".join(x.to_bytes(32, "big") for x in v), tb[:32 * m], tb[32 * m:])
            assert verify(gens, n, m, proof, coms, 2) == 3
            assert verify(gens, n, m, proof, coms, 0) == 3
        finally:
            gens.close()


def test_group_primitives(ctx, oracle_c):
    rng = random.Random(77)
    n = 37
    pts = oracle_c.mul_batch([ec.G] * (2 * n), [rng.randrange(1, ec.Q) for _ in range(2 * n)])
    ks = [rng.randrange(ec.Q) for _ in range(n)]
    ks[0], ks[1], ks[2] = 0, 1, ec.Q - 1
    pb = b"".join(ec.enc64(p) for p in pts)
    got = ctx.scalar_mul_batch(pb[:64 * n], b"".join(k.to_bytes(32, "big") for k in ks))
    assert got == b"".join(ec.enc64(p) for p in oracle_c.mul_batch(pts[:n], ks))
    s_lo, s_hi = rng.randrange(1, ec.Q), rng.randrange(1, ec.Q)
    got = ctx.fold_bases(pb[:64 * n], pb[64 * n:], s_lo.to_bytes(32, "big"), s_hi.to_bytes(32, "big"))
    want = [oracle_c.add(oracle_c.mul(pts[i], s_lo), oracle_c.mul(pts[n + i], s_hi)) for i in range(n)]
    assert got == b"".join(ec.enc64(p) for p in want)
    # duplicates / opposite points fold to 2P-ish and infinity correctly
    got = ctx.fold_bases(pb[:64], ec.enc64(ec.neg(pts[0])), (5).to_bytes(32, "big"), (5).to_bytes(32, "big"))
    assert got == b"\0" * 64
    got = ctx.fixed_base_mul_batch(b"".join(k.to_bytes(32, "big") for k in ks))
    assert got == b"".join(ec.enc64(p) for p in oracle_c.mul_batch([ec.G] * n, ks))


def test_bpplus_c4_n64_m64(ctx):
    """BASELINE.json configs[3]: Bulletproofs+ aggregated n=64 m=64 prove + verify (MSM of 8284 terms)."""
    case = load("bpplus_c4_n64_m64.json")
    gens = _gens(ctx, 1, 64, 64, 4)
    try:
        proof = ctx.wip_rangeproof_prove(gens, 64, _be(case["secrets"]), _be(case["blindings"]), bytes.fromhex(case["tape"]))
        assert len(proof) == 987 and proof.hex() == case["proof"]
        items = [(case["proof"], case["commitments"], case["verdict"])] + [(t["proof"], t["commitments"], t["verdict"]) for t in case["tampered"]]
        for pf, cm, want in items:
            coms = b"".join(bytes.fromhex(c) for c in cm)
            assert ctx.wip_rangeproof_verify(gens, 64, 64, bytes.fromhex(pf), coms, 2) == VERDICT_CODE[want]
        coms = b"".join(bytes.fromhex(c) for c in case["commitments"])
        assert ctx.wip_rangeproof_verify(gens, 64, 64, proof, coms, 0) == 0
    finally:
        gens.close()


@pytest.mark.parametrize("kind,seed,n,m", [(0, None, 64, 1), (1, None, 64, 1), (0, None, 8, 1), (1, None, 32, 4), (0, 1, 3, 5),
                                           (1, 1, 1, 1), (0, (1 << 64) - 40, 16, 2), (1, None, 64, 64)])
def test_generate_bases_on_device_matches_host_and_oracle(ctx, kind, seed, n, m):
    """generate_random_point / StatementRP::generate_bases (range_proof.rs:785-808, 821-843;
    range_proof_wip.rs:48-91) on the GPU: byte-identical to the host recipe for every label (about half of them take
    the try-and-double path, ~1/256 lose a leading zero byte of the SHA-512 digest, labels of 1..8 bytes, label
    wrap-around at 2^64), and to the oracle's Python restatement on the small cases."""
    from oracle import curv
    seed = bpa.KZEN_SEED if seed is None else seed
    want = bpa.generate_bases(kind, seed, n, m)
    got = ctx.generate_bases(kind, seed, n, m)
    assert got == want
    if n * m <= 64 and seed + 2 * n * m < (1 << 64):
        def lab(x):
            return ec.enc64(curv.label_point(x))
        nm = n * m
        assert got[0][:64 * min(nm, 8)] == b"".join(lab(seed + i) for i in range(min(nm, 8)))
        assert got[1][:64 * min(nm, 8)] == b"".join(lab(seed + n + i) for i in range(min(nm, 8)))
        assert got[3] == lab(1 if kind == 0 else seed - 1)


@pytest.mark.parametrize("kind,n,m,count", [(0, 64, 1, 40), (0, 16, 2, 7), (1, 64, 1, 33), (1, 8, 4, 5), (0, 64, 1, 150)])
def test_prove_batch_matches_single_proofs(ctx, kind, n, m, count):
    """bp_rangeproof_prove_batch / bp_wip_rangeproof_prove_batch (the device batch provers: per-proof vectors as kernels over
    [proof][i], one table-MSM launch per protocol step for all proofs, the transcript on the host): every proof byte-identical
    to the single-proof call on the same inputs (whose bytes the tests above pin to the oracle), batches that span both
    lanes, and all of them verify."""
    from oracle import tape
    g, h, G, H = bpa.generate_bases(kind, bpa.KZEN_SEED, n, m)
    gens = bpa.Gens(ctx, g, h, G, H, 8)
    try:
        nm = n * m
        k = nm.bit_length() - 1
        ntape = (1 + 2 * k + 4) if kind else (2 + 2 * nm + 2)
        rng = random.Random(1000 * kind + n + m)
        secs, blinds, tapes = b"", b"", b""
        for i in range(count):
            tb = tape.make_tape_bytes("pb-%d-%d-%d-%d" % (kind, n, m, i), m + ntape)
            secs += b"".join(rng.randrange(1 << n).to_bytes(32, "big") for _ in range(m))
            blinds += tb[:32 * m]
            tapes += tb[32 * m:]
        batch = (ctx.wip_rangeproof_prove_batch if kind else ctx.rangeproof_prove_batch)(gens, n, m, secs, blinds, tapes, 32 * ntape)
        single = ctx.wip_rangeproof_prove if kind else ctx.rangeproof_prove
        assert len(batch) == count
        step = 1 if count <= 40 else 9
        for i in range(0, count, step):
            want = single(gens, n, secs[32 * m * i:32 * m * (i + 1)], blinds[32 * m * i:32 * m * (i + 1)], tapes[32 * ntape * i:32 * ntape * (i + 1)])
            assert batch[i] == want, i
        # all of them verify against commitments made on the GPU (V_j = v_j G + r_j H)
        coms = b"".join(ctx.msm(G + H, secs[32 * j:32 * j + 32] + blinds[32 * j:32 * j + 32], None) for j in range(count * m))
        verify = ctx.wip_rangeproof_verify_batch if kind else ctx.rangeproof_verify_batch
        assert verify(gens, n, m, b"".join(batch), coms) == [0] * count
    finally:
        gens.close()


def test_prove_batch_with_caller_G_and_one_bad_tape(ctx, oracle_c):
    """The device batch prover with G != Point::generator() (y*G uses the table's extra generator row, x_u*G the caller's G)
    and one proof of the batch with a zero tape scalar: that proof reports BP_EINVAL, the others are byte-identical to the
    single-proof call."""
    n, m, count = 16, 2, 9
    g, h, _, H = bpa.generate_bases(0, bpa.KZEN_SEED, n, m)
    G = ec.enc64(oracle_c.mul(ec.G, 0xC0FFEE1234567))
    gens = bpa.Gens(ctx, g, h, G, H, 6)
    try:
        from oracle import tape
        ntape = 2 + 2 * n * m + 2
        secs, blinds, tapes = b"", b"", b""
        for i in range(count):
            tb = tape.make_tape_bytes("pbG-%d" % i, m + ntape)
            secs += b"".join(((i * 977 + j) % (1 << n)).to_bytes(32, "big") for j in range(m))
            blinds += tb[:32 * m]
            tapes += tb[32 * m:]
        batch = ctx.rangeproof_prove_batch(gens, n, m, secs, blinds, tapes, 32 * ntape)
        for i in range(count):
            want = ctx.rangeproof_prove(gens, n, secs[64 * i:64 * i + 64], blinds[64 * i:64 * i + 64], tapes[32 * ntape * i:32 * ntape * (i + 1)])
            assert batch[i] == want, i
        bad = bytearray(tapes)
        bad[32 * ntape * 4 + 32 * 7:32 * ntape * 4 + 32 * 8] = b"\0" * 32
        with pytest.raises(bpa.BpError) as e:
            ctx.rangeproof_prove_batch(gens, n, m, secs, blinds, bytes(bad), 32 * ntape)
        assert e.value.code == -1
    finally:
        gens.close()


def test_prove_batch_reports_bad_inputs(ctx):
    g, h, G, H = bpa.generate_bases(0, bpa.KZEN_SEED, 8, 1)
    gens = bpa.Gens(ctx, g, h, G, H, 4)
    try:
        with pytest.raises(bpa.BpError) as e:  # zero scalars on the tape: the reference's Scalar::random() never yields them
            ctx.rangeproof_prove_batch(gens, 8, 1, (5).to_bytes(32, "big") * 3, (7).to_bytes(32, "big") * 3, b"\0" * (32 * 20 * 3), 32 * 20)
        assert e.value.code == -1
        ctx.set_proto_path(1)
        try:
            with pytest.raises(bpa.BpError):
                ctx.rangeproof_prove_batch(gens, 8, 1, (5).to_bytes(32, "big"), (7).to_bytes(32, "big"), b"\x01" * (32 * 20), 32 * 20)
        finally:
            ctx.set_proto_path(0)
    finally:
        gens.close()


@pytest.mark.parametrize("kind", (0, 1))
def test_caller_G_different_from_the_curve_generator(ctx, oracle_c, kind):
    """y*G of the z challenge multiplies Point::generator() (range_proof.rs:125-126, 570-571; range_proof_wip.rs:169,
    422), everything else the caller's G: with G != generator the proof bytes and every verifier's verdict still match
    the oracle (bp_gens keeps an extra table row for the generator in that case)."""
    from oracle import curv, encoding, tape
    from oracle.range_proof import RangeProof
    from oracle.range_proof_wip import RangeProofWIP, StatementRP
    n, m = 8, 2
    g_vec, h_vec, _, H = (curv.bpplus_generators if kind else curv.bp_generators)(n, m)
    G = oracle_c.mul(ec.G, 0xC0FFEE1234567)
    tb = tape.make_tape_bytes("ownG-%d" % kind, m + (tape.bpplus_tape_len if kind else tape.bp_tape_len)(n, m))
    v = [200, 17]
    r = [int.from_bytes(tb[32 * j:32 * j + 32], "big") for j in range(m)]
    com = [oracle_c.add(oracle_c.mul(G, v[j]), oracle_c.mul(H, r[j])) for j in range(m)]
    if kind:
        stmt = StatementRP(g_vec, h_vec, G, H, n)
        want = RangeProofWIP.prove(stmt, v, r, tape.Tape(tb[32 * m:]))
        assert want.aggregated_verify(stmt, com) == "Ok"
        want_b = encoding.encode_bpplus(want)
    else:
        want = RangeProof.prove(g_vec, h_vec, G, H, v, r, n, tape.Tape(tb[32 * m:]))
        assert want.aggregated_verify(g_vec, h_vec, G, H, com, n) == "Ok"
        want_b = encoding.encode_bp(want)
    enc = lambda ps: b"".join(ec.enc64(p) for p in ps)
    gens = bpa.Gens(ctx, enc(g_vec), enc(h_vec), ec.enc64(G), ec.enc64(H), 5)
    try:
        prove = ctx.wip_rangeproof_prove if kind else ctx.rangeproof_prove
        got = prove(gens, n, b"".join(x.to_bytes(32, "big") for x in v), tb[:32 * m], tb[32 * m:])
        assert got == want_b
        coms = enc(com)
        bad = bytearray(got)
        bad[-1] ^= 1
        single = ctx.wip_rangeproof_verify if kind else ctx.rangeproof_verify
        for mode in ((0, 2) if kind else (0, 1, 2)):
            assert single(gens, n, m, got, coms, mode) == 0, mode
            assert single(gens, n, m, bytes(bad), coms, mode) != 0, mode
        batch = ctx.wip_rangeproof_verify_batch if kind else ctx.rangeproof_verify_batch
        assert batch(gens, n, m, got + bytes(bad) + got, coms * 3) == [0, 3, 0]
        rlc = ctx.wip_rangeproof_verify_batch_rlc if kind else ctx.rangeproof_verify_batch_rlc
        assert rlc(gens, n, m, got * 3, coms * 3, bytes(range(32))) == ([0, 0, 0], False)
        ctx.set_proto_path(1)
        try:
            assert prove(gens, n, b"".join(x.to_bytes(32, "big") for x in v), tb[:32 * m], tb[32 * m:]) == want_b
        finally:
            ctx.set_proto_path(0)
    finally:
        gens.close()


def test_single_proof_calls_from_several_threads(ctx):
    """The single-proof calls' host side (round 3): window tables of the proof's points on the side stream, scalars and
    decompressions on the process-wide host workers.  Four contexts on four threads prove and verify the n=64 m=16 fixture
    (n*m = 1024: the size from which the workers are used; a second caller's pass runs inline) at once; bytes and verdicts
    as the fixture's; then the first context's calls again in a different order, a batch call between them (its workspaces and
    the side stream are shared with the single-proof path)."""
    import threading

    case = load("bp_c2_n64_m16.json")
    n, m = case["n"], case["m"]
    g, h, G, H = bpa.generate_bases(0, bpa.KZEN_SEED, n, m)
    proof, coms = bytes.fromhex(case["proof"]), b"".join(bytes.fromhex(c) for c in case["commitments"])
    sec, bl, tape = _be(case["secrets"]), _be(case["blindings"]), bytes.fromhex(case["tape"])
    bad = bytearray(proof)
    bad[-1] ^= 1  # b of the inner-product argument: RangeProofError
    ctxs = [ctx] + [bpa.Context(0) for _ in range(3)]
    gens = [bpa.Gens(c, g, h, G, H, 8) for c in ctxs]
    errs = []

    def work(i):
        try:
            c, gn = ctxs[i], gens[i]
            for rep in range(3):
                assert c.rangeproof_verify(gn, n, m, proof, coms, 2) == VERDICT_CODE["Ok"]
                assert c.rangeproof_verify(gn, n, m, bytes(bad), coms, 2) == VERDICT_CODE["RangeProofError"]
                if (i + rep) % 2 == 0:
                    assert c.rangeproof_prove(gn, n, sec, bl, tape).hex() == case["proof"]
                assert c.rangeproof_verify(gn, n, m, proof, coms, 1) == VERDICT_CODE["Ok"]
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errs.append((i, repr(e)))

    thr = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in thr:
        t.start()
    for t in thr:
        t.join()
    try:
        assert not errs, errs
        assert ctx.rangeproof_verify(gens[0], n, m, bytes(bad), coms, 2) == VERDICT_CODE["RangeProofError"]
        v = ctx.rangeproof_verify_batch(gens[0], n, m, proof + bytes(bad) + proof, coms * 3)
        assert list(v) == [VERDICT_CODE["Ok"], VERDICT_CODE["RangeProofError"], VERDICT_CODE["Ok"]]
        assert ctx.rangeproof_verify(gens[0], n, m, proof, coms, 2) == VERDICT_CODE["Ok"]
    finally:
        for gn in gens:
            gn.close()
        for c in ctxs[1:]:
            c.close()
