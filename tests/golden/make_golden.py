#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ with the CPU oracle (oracle/).

The reference (Rust, curv-kzen 0.10 not vendored, no cargo here) cannot be run, and its own tests
hold no byte-level vectors, so these vectors come from the oracle restatement:
they pin the oracle against regressions and give the GPU tests / bench.py fixed inputs.  Vectors
the reference itself pins (test_wip = 46, curv DigestExt doc digest) are asserted here too.

Usage:  python tests/golden/make_golden.py            (writes *.json next to this file)
"""
import hashlib
import json
import multiprocessing as mp
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import curv, ec, encoding, tape  # noqa: E402
from oracle.range_proof import RangeProof  # noqa: E402
from oracle.range_proof_wip import RangeProofWIP, StatementRP  # noqa: E402
from oracle.weighted_inner_product import weighted_inner_product  # noqa: E402


def hx(b):
    return bytes(b).hex()


def secrets_for(seed, n, m):
    return [int.from_bytes(hashlib.sha256(("secret-%s-%d" % (seed, j)).encode()).digest(), "big") % (1 << n) for j in range(m)]


def make_bp(n, m, seed, with_trace=True):
    g_vec, h_vec, G, H = curv.bp_generators(n, m)
    tb = tape.make_tape_bytes("bpmsm-seed-%s" % seed, m + tape.bp_tape_len(n, m))
    tp = tape.Tape(tb)
    v = secrets_for(seed, n, m)
    r = [tp.random_scalar() for _ in range(m)]
    com = [ec.add(ec.mul(G, v[j]), ec.mul(H, r[j])) for j in range(m)]
    tr = {} if with_trace else None
    proof = RangeProof.prove(g_vec, h_vec, G, H, v, r, n, tp, trace=tr)
    assert tp.pos == len(tb)
    enc = encoding.encode_bp(proof)
    out = {
        "kind": "bp", "n": n, "m": m, "seed": seed,
        "secrets": [hex(x) for x in v],
        "blindings": [hex(x) for x in r],
        "tape": hx(tb[32 * m:]),
        "commitments": [hx(ec.enc64(c)) for c in com],
        "proof": hx(enc),
    }
    if with_trace:
        vt = {}
        init, points, scalars = proof.aggregated_verify_terms(g_vec, h_vec, G, H, com, n, trace=vt)
        out["challenges"] = {k: hex(vt[k]) for k in ("y", "z", "x", "x_u", "c")}
        out["challenges"]["xs"] = [hex(x) for x in vt["xs"]]
        out["msm_init"] = hx(ec.enc64(init))
        out["msm_scalars_sha256"] = hashlib.sha256(b"".join(s.to_bytes(32, "big") for s in scalars)).hexdigest()
        assert ec.msm_naive(points, scalars, init) == proof.A
    return out, proof, (g_vec, h_vec, G, H), com


def tamper_cases_bp(proof_bytes, coms, k):
    """(name, proof, coms) variants: SURVEY.md §8d C3 tampered subset."""
    out = []
    b = bytearray(proof_bytes)
    t = bytearray(b); t[4 * 33 + 64 + 31] ^= 1; out.append(("flip_tx_bit", bytes(t), coms))
    t = bytearray(b); t[4 * 33 + 96 + 66 * k + 31] ^= 1; out.append(("flip_a_tag_bit", bytes(t), coms))
    t = bytearray(b); t[4 * 33 + 31] ^= 0x80; out.append(("flip_tau_x_bit", bytes(t), coms))
    # swap L_1 and R_1 (still valid points, wrong statement)
    t = bytearray(b)
    o = 4 * 33 + 96
    t[o:o + 33], t[o + 33 * k:o + 33 * k + 33] = b[o + 33 * k:o + 33 * k + 33], b[o:o + 33]
    out.append(("swap_L1_R1", bytes(t), coms))
    # replace T1 by another valid point (2*T1 is on the curve)
    T1 = ec.decompress(bytes(b[66:99]))
    t = bytearray(b); t[66:99] = ec.compress(ec.add(T1, T1)); out.append(("T1_doubled", bytes(t), coms))
    # wrong commitment
    c2 = list(coms); c2[0] = ec.add(c2[0], ec.G); out.append(("commitment_plus_G", bytes(b), c2))
    # malformed: x coordinate not on curve / scalar >= q
    t = bytearray(b)
    x = int.from_bytes(t[1:33], "big")
    while True:
        x = (x + 1) % ec.P
        try:
            ec.decompress(bytes([2]) + x.to_bytes(32, "big"))
        except ValueError:
            break
    t[1:33] = x.to_bytes(32, "big"); out.append(("A_not_on_curve", bytes(t), coms))
    t = bytearray(b); t[4 * 33 + 64:4 * 33 + 96] = (ec.Q + 1).to_bytes(32, "big"); out.append(("tx_ge_q", bytes(t), coms))
    return out


def bp_case(n, m, seed):
    out, proof, gens, com = make_bp(n, m, seed)
    g_vec, h_vec, G, H = gens
    k = (n * m).bit_length() - 1
    assert proof.verify(g_vec, h_vec, G, H, com, n) == "Ok"
    assert proof.fast_verify(g_vec, h_vec, G, H, com, n) == "Ok"
    assert proof.aggregated_verify(g_vec, h_vec, G, H, com, n) == "Ok"
    out["verdict"] = "Ok"
    tampered = []
    for name, pb, cc in tamper_cases_bp(bytes.fromhex(out["proof"]), com, k):
        try:
            p2 = encoding.decode_bp(pb)
            if any(s >= ec.Q for s in (p2.tau_x, p2.miu, p2.tx, p2.inner_product_proof.a_tag, p2.inner_product_proof.b_tag)):
                raise ValueError("scalar >= q")
            verdict = p2.aggregated_verify(g_vec, h_vec, G, H, cc, n)
            if n * m <= 128:  # equivalence triangle on small cases
                assert p2.verify(g_vec, h_vec, G, H, cc, n) == verdict == p2.fast_verify(g_vec, h_vec, G, H, cc, n)
        except ValueError:
            verdict = "Malformed"
        tampered.append({"name": name, "proof": hx(pb), "commitments": [hx(ec.enc64(c)) for c in cc], "verdict": verdict})
    out["tampered"] = tampered
    return out


def _batch_one(i):
    out, _, _, _ = make_bp(64, 1, "c3-%d" % i, with_trace=False)
    return out["proof"], out["commitments"][0]


def make_bpplus(n, m, seed):
    stmt = StatementRP.generate_bases(curv.KZEN_SEED, m, n)
    tb = tape.make_tape_bytes("bpmsm-seed-plus-%s" % seed, m + tape.bpplus_tape_len(n, m))
    tp = tape.Tape(tb)
    v = secrets_for(seed, n, m)
    r = [tp.random_scalar() for _ in range(m)]
    com = [ec.add(ec.mul(stmt.G, v[j]), ec.mul(stmt.H, r[j])) for j in range(m)]
    proof = RangeProofWIP.prove(stmt, v, r, tp)
    assert tp.pos == len(tb)
    vt = {}
    init, points, scalars = proof.aggregated_verify_terms(stmt, com, trace=vt)
    assert ec.msm_naive(points, scalars, init) == proof.weighted_inner_product_proof.b_tag
    enc = encoding.encode_bpplus(proof)
    out = {
        "kind": "bp+", "n": n, "m": m, "seed": seed,
        "secrets": [hex(x) for x in v], "blindings": [hex(x) for x in r],
        "tape": hx(tb[32 * m:]), "commitments": [hx(ec.enc64(c)) for c in com], "proof": hx(enc),
        "challenges": {"y": hex(vt["y"]), "z": hex(vt["z"]), "e": hex(vt["e"]), "xs": [hex(x) for x in vt["xs"]]},
        "msm_init": hx(ec.enc64(init)),
        "msm_scalars_sha256": hashlib.sha256(b"".join(s.to_bytes(32, "big") for s in scalars)).hexdigest(),
        "verdict": "Ok",
    }
    if n * m <= 256:
        assert proof.verify(stmt, com) == "Ok"
    k = (n * m).bit_length() - 1
    b = bytes.fromhex(out["proof"])
    tam = []
    t = bytearray(b); t[-1] ^= 1; tam.append(("flip_delta_prime_bit", bytes(t), com))
    t = bytearray(b); t[33 + 66 * k + 66 + 31] ^= 1; tam.append(("flip_r_prime_bit", bytes(t), com))
    c2 = list(com); c2[-1] = ec.add(c2[-1], ec.G); tam.append(("commitment_plus_G", b, c2))
    # swap L_1 and R_1 / a_tag and b_tag (still valid points, wrong statement)
    t = bytearray(b); t[33:66], t[33 + 33 * k:66 + 33 * k] = b[33 + 33 * k:66 + 33 * k], b[33:66]; tam.append(("swap_L1_R1", bytes(t), com))
    o = 33 + 66 * k
    t = bytearray(b); t[o:o + 33], t[o + 33:o + 66] = b[o + 33:o + 66], b[o:o + 33]; tam.append(("swap_a_tag_b_tag", bytes(t), com))
    # malformed: A's x coordinate not on the curve / scalar >= q
    t = bytearray(b)
    x = int.from_bytes(t[1:33], "big")
    while True:
        x = (x + 1) % ec.P
        try:
            ec.decompress(bytes([2]) + x.to_bytes(32, "big"))
        except ValueError:
            break
    t[1:33] = x.to_bytes(32, "big"); tam.append(("A_not_on_curve", bytes(t), com))
    t = bytearray(b); t[o + 66:o + 98] = (ec.Q + 5).to_bytes(32, "big"); tam.append(("r_prime_ge_q", bytes(t), com))
    t = bytearray(b); t[o + 33] = 4; tam.append(("b_tag_bad_prefix", bytes(t), com))
    tampered = []
    for name, pb, cc in tam:
        try:
            p2 = encoding.decode_bpplus(pb)
            w = p2.weighted_inner_product_proof
            if any(s >= ec.Q for s in (w.r_prime, w.s_prime, w.delta_prime)):
                raise ValueError("scalar >= q")
            verdict = p2.aggregated_verify(stmt, cc)
            if n * m <= 128:
                assert p2.verify(stmt, cc) == verdict
        except ValueError:
            verdict = "Malformed"
        tampered.append({"name": name, "proof": hx(pb), "commitments": [hx(ec.enc64(c)) for c in cc], "verdict": verdict})
    out["tampered"] = tampered
    return out


def main():
    assert ec.have_c(), "build the C oracle first: make -C oracle"
    # reference-pinned known answers
    assert weighted_inner_product([3, 2, 1, 0], [1, 2, 3, 4], 2) == 46  # weighted_inner_product.rs:828-861
    kats = {
        "curv_digest_ext_doc_example": "73764f937fbe25092466b417fa66ad9c62607865e1f8151df253aa3a2fd7599b",
        "test_wip": 46,
        "two_G_x": "c6047f9441ed7d6d3045406e95c07cd85c778e4b8cef3ca7abac09b95c709ee5",
    }
    with open(os.path.join(HERE, "kats.json"), "w") as f:
        json.dump(kats, f, indent=1)

    plus_only = "--plus-only" in sys.argv  # regenerate only the Bulletproofs+ fixtures (same seeds: same proofs)
    cases = []
    for (n, m, seed) in [] if plus_only else [(8, 1, "t8x1"), (16, 2, "t16x2"), (32, 4, "t32x4"), (64, 1, "c1"), (64, 4, "t64x4")]:
        print("bp", n, m, flush=True)
        cases.append(bp_case(n, m, seed))
    if not plus_only:
        with open(os.path.join(HERE, "bp_cases.json"), "w") as f:
            json.dump(cases, f)

        print("bp 64x16 (config C2)", flush=True)
        with open(os.path.join(HERE, "bp_c2_n64_m16.json"), "w") as f:
            json.dump(bp_case(64, 16, "c2"), f)

    print("bp+ cases", flush=True)
    plus = [make_bpplus(n, m, s) for (n, m, s) in [(8, 1, "p8x1"), (16, 2, "p16x2"), (32, 4, "p32x4"), (64, 4, "p64x4")]]
    with open(os.path.join(HERE, "bpplus_cases.json"), "w") as f:
        json.dump(plus, f)

    print("bp+ 64x64 (config C4)", flush=True)
    with open(os.path.join(HERE, "bpplus_c4_n64_m64.json"), "w") as f:
        json.dump(make_bpplus(64, 64, "c4"), f)

    if plus_only:
        print("done (plus only)")
        return
    # C3 batch: 256 distinct n=64 m=1 proofs with the shared generators (bench.py tiles them)
    print("c3 batch", flush=True)
    with mp.Pool(8) as pool:
        res = pool.map(_batch_one, range(256))
    g_vec, h_vec, G, H = curv.bp_generators(64, 1)
    batch = {
        "kind": "bp", "n": 64, "m": 1, "count": len(res),
        "g_vec": hx(b"".join(ec.enc64(p) for p in g_vec)), "h_vec": hx(b"".join(ec.enc64(p) for p in h_vec)),
        "G": hx(ec.enc64(G)), "H": hx(ec.enc64(H)),
        "proofs": "".join(p for p, _ in res), "commitments": "".join(c for _, c in res),
    }
    with open(os.path.join(HERE, "bp_c3_batch_n64_m1.json"), "w") as f:
        json.dump(batch, f)
    print("done")


if __name__ == "__main__":
    main()
