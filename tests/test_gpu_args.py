"""GPU parity of the standalone argument entry points (bp_ipa_prove / bp_ipa_verify, bp_wipa_prove / bp_wipa_verify)
through the C ABI: the reference's own argument tests — make_ipp_{32..1}(+_fast_verify), make_ipp_non_power_2
(inner_product.rs:602-690), make_wip_{32..1}(+_fast_verify), make_wip_non_power_2 (weighted_inner_product.rs:863-948) —
with the argument BYTES compared with the oracle's restatement for the same a, b (and randomness tape), both verifiers
(recursive and single-MSM) on the honest argument, on a wrong P and on tampered arguments."""
import random

import pytest

import bulletproofs_amd as bpa
from oracle import curv, ec, encoding, tape
from oracle.inner_product import InnerProductArg, inner_product
from oracle.weighted_inner_product import WeightedInnerProdArg, weighted_inner_product

pytestmark = pytest.mark.gpu


def _enc(ps):
    return b"".join(ec.enc64(p) for p in ps)


def _be(xs):
    return b"".join((x % ec.Q).to_bytes(32, "big") for x in xs)


def _ipa_inputs(n, pad_to=None):
    """inner_product.rs:373-451 (test_helper) / :533-600 (non power of two: a, b zero-padded, bases for the padded length)"""
    N = pad_to or n
    rng = random.Random(1000 + n)
    g_vec = [curv.label_point(i + curv.KZEN_SEED) for i in range(N)]
    h_vec = [curv.label_point(N + i + curv.KZEN_SEED) for i in range(N)]
    Gx = curv.label_point(1)
    a = [rng.randrange(1, ec.Q) for _ in range(n)] + [0] * (N - n)
    b = [rng.randrange(1, ec.Q) for _ in range(n)] + [0] * (N - n)
    y = rng.randrange(1, ec.Q)
    hi_tag = ec.mul_batch(h_vec, [pow(pow(y, i, ec.Q), -1, ec.Q) for i in range(N)])
    P = ec.msm_naive(hi_tag, b, ec.msm_naive(g_vec, a, ec.mul(Gx, inner_product(a, b))))
    return g_vec, hi_tag, Gx, P, a, b


@pytest.mark.parametrize("n,pad", [(1, None), (2, None), (4, None), (8, None), (16, None), (32, None), (9, 16)])
def test_ipa_prove_and_both_verifiers(ctx, oracle_c, n, pad):
    g_vec, hi_tag, Gx, P, a, b = _ipa_inputs(n, pad)
    want = InnerProductArg.prove(g_vec, hi_tag, Gx, P, a, b)
    want_b = encoding.encode_ipa(want)
    gb, hb, ub, Pb = _enc(g_vec), _enc(hi_tag), ec.enc64(Gx), ec.enc64(P)
    got = ctx.ipa_prove(gb, hb, ub, _be(a), _be(b), Pb)
    assert got == want_b
    bad_P = ec.enc64(ec.add(P, ec.G))
    for mode in (0, 1):
        assert ctx.ipa_verify(gb, hb, ub, Pb, got, mode) == 0
        assert ctx.ipa_verify(gb, hb, ub, bad_P, got, mode) == 1      # Err(InnerProductError)
        t = bytearray(got)
        t[-1] ^= 1                                                     # b_tag
        v = 1 if int.from_bytes(t[-32:], "big") < ec.Q else 255
        assert ctx.ipa_verify(gb, hb, ub, Pb, bytes(t), mode) == v
        if len(want.L) >= 2:
            k = len(want.L)
            t = bytearray(got)
            t[0:33], t[33:66] = got[33:66], got[0:33]                  # L_1 <-> L_2
            w2 = InnerProductArg([want.L[1], want.L[0]] + want.L[2:], want.R, want.a_tag, want.b_tag)
            exp = (w2.verify if mode == 0 else w2.fast_verify)(g_vec, hi_tag, Gx, P)
            assert exp == "InnerProductError"
            assert ctx.ipa_verify(gb, hb, ub, Pb, bytes(t), mode) == 1
            t = bytearray(got)
            t[33 * k] = 7                                              # R_1: not a SEC1 prefix
            assert ctx.ipa_verify(gb, hb, ub, Pb, bytes(t), mode) == 255


def test_ipa_argument_checks(ctx):
    g_vec, hi_tag, Gx, P, a, b = _ipa_inputs(4)
    gb, hb, ub, Pb = _enc(g_vec), _enc(hi_tag), ec.enc64(Gx), ec.enc64(P)
    with pytest.raises(bpa.BpError) as e:          # assert!(n.is_power_of_two())  inner_product.rs:53
        ctx.ipa_prove(gb[:64 * 3], hb[:64 * 3], ub, _be(a[:3]), _be(b[:3]))
    assert e.value.code == bpa.BP_EINVAL
    arg = ctx.ipa_prove(gb, hb, ub, _be(a), _be(b))
    with pytest.raises(bpa.BpError):               # wrong argument length for n
        ctx.ipa_verify(gb, hb, ub, Pb, arg[:-1], 0)
    off = bytearray(ub)
    off[63] ^= 1
    with pytest.raises(bpa.BpError) as e:          # ux not on the curve: Point::from_bytes would fail
        ctx.ipa_verify(gb, hb, bytes(off), Pb, arg, 1)
    assert e.value.code == -2


def _wip_inputs(n, pad_to=None):
    """weighted_inner_product.rs:538-613 (test_helper) / :715-825 (non power of two)"""
    N = pad_to or n
    rng = random.Random(2000 + n)
    g_vec = [curv.label_point(i + curv.KZEN_SEED) for i in range(N)]
    h_vec = [curv.label_point(N + i + curv.KZEN_SEED) for i in range(N)]
    g, h = curv.label_point(2), curv.label_point(3)
    a = [rng.randrange(1, ec.Q) for _ in range(n)] + [0] * (N - n)
    b = [rng.randrange(1, ec.Q) for _ in range(n)] + [0] * (N - n)
    y = curv.sha256().chain(b"Seed string decided by P,V!").result_bigint() % ec.Q
    alpha = rng.randrange(1, ec.Q)
    P = ec.msm_naive(h_vec, b, ec.msm_naive(g_vec, a, ec.add(ec.mul(g, weighted_inner_product(a, b, y)), ec.mul(h, alpha))))
    tb = tape.make_tape_bytes("wipa-%d" % n, 2 * (N.bit_length() - 1) + 4)
    return g_vec, h_vec, g, h, P, a, b, alpha, y, tb


@pytest.mark.parametrize("n,pad", [(1, None), (2, None), (4, None), (8, None), (16, None), (32, None), (9, 16)])
def test_wipa_prove_and_both_verifiers(ctx, oracle_c, n, pad):
    g_vec, h_vec, g, h, P, a, b, alpha, y, tb = _wip_inputs(n, pad)
    want = WeightedInnerProdArg.prove(g_vec, h_vec, g, h, P, a, b, alpha, y, tape.Tape(tb))
    want_b = encoding.encode_wipa(want)
    gb, hb, gp, hp, Pb, yb = _enc(g_vec), _enc(h_vec), ec.enc64(g), ec.enc64(h), ec.enc64(P), y.to_bytes(32, "big")
    got = ctx.wipa_prove(gb, hb, gp, hp, _be(a), _be(b), alpha.to_bytes(32, "big"), yb, tb, Pb)
    assert got == want_b
    bad_P = ec.enc64(ec.add(P, ec.G))
    k = len(want.L)
    for mode in (0, 1):
        assert ctx.wipa_verify(gb, hb, gp, hp, Pb, yb, got, mode) == 0
        assert ctx.wipa_verify(gb, hb, gp, hp, bad_P, yb, got, mode) == 2   # Err(WeightedInnerProdError)
        y2 = ((y + 1) % ec.Q).to_bytes(32, "big")
        assert ctx.wipa_verify(gb, hb, gp, hp, Pb, y2, got, mode) == 2
        t = bytearray(got)
        t[-33] ^= 1                                                          # s'
        v = 2 if int.from_bytes(t[-64:-32], "big") < ec.Q else 255
        assert ctx.wipa_verify(gb, hb, gp, hp, Pb, yb, bytes(t), mode) == v
        t = bytearray(got)
        t[66 * k:66 * k + 33], t[66 * k + 33:66 * k + 66] = got[66 * k + 33:66 * k + 66], got[66 * k:66 * k + 33]   # a_tag <-> b_tag
        assert ctx.wipa_verify(gb, hb, gp, hp, Pb, yb, bytes(t), mode) == 2
        assert ctx.wipa_verify(gb, hb, gp, hp, Pb, b"\0" * 32, got, mode) == 255   # y = 0: mod_inv(y).unwrap() panics


def test_wipa_matches_the_range_proof_path(ctx, oracle_c):
    """The argument inside a Bulletproofs+ range proof is the same object: the tail of a golden proof verifies through
    bp_wipa_verify against A_hat computed as RangeProofWIP::verify does (range_proof_wip.rs:352-376)."""
    from _golden import bpplus_cases
    from oracle import range_proof_wip as rpw
    case = bpplus_cases()[0]
    n, m = case["n"], case["m"]
    nm = n * m
    g_vec, h_vec, G, H = curv.bpplus_generators(n, m)
    proof = encoding.decode_bpplus(bytes.fromhex(case["proof"]))
    com = [ec.dec64(bytes.fromhex(c)) for c in case["commitments"]]
    y = curv.sha256().chain_points([proof.A]).result_scalar()
    z = curv.sha256().chain_points([proof.A, ec.mul(ec.G, y)]).result_scalar()
    y_powers, y_powers_rev, vec_z2m, d, y_pow, scalar_G = rpw._common_scalars(n, m, y, z)
    sum_com = ec.msm_naive(com, [y_pow * vec_z2m[i] % ec.Q for i in range(m)], proof.A)
    A_hat = ec.msm_naive(list(g_vec) + list(h_vec) + [G], [(0 - z) % ec.Q] * nm + [(d[i] * y_powers_rev[i] + z) % ec.Q for i in range(nm)] + [scalar_G],
                         sum_com)
    arg = bytes.fromhex(case["proof"])[33:]
    for mode in (0, 1):
        assert ctx.wipa_verify(_enc(g_vec), _enc(h_vec), ec.enc64(G), ec.enc64(H), ec.enc64(A_hat), y.to_bytes(32, "big"), arg, mode) == 0
