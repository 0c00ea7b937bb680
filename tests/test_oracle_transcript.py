"""Pins the transcript byte format and the scalar known-answer test the reference holds."""
import hashlib

from oracle import curv, ec
from oracle.weighted_inner_product import weighted_inner_product

from _golden import load


def test_curv_digest_ext_doc_example():
    """curv DigestExt doc example: chain_point = 65-byte uncompressed,
    chain_bigint = minimal big-endian, result_bigint = big-endian digest."""
    bp2 = (0x08d13221e3a7326a34dd45214ba80116dd142e4b5ff3ce66a8dc7bfa0378b795,
           0x5d41ac1477614b5c0848d50dbd565ea2807bcba1df0df07a8217e9f7f7c2be88)
    assert ec.on_curve(bp2)
    d = curv.sha256().chain_point(ec.G).chain_point(bp2).chain_bigint(10).result_bigint()
    assert "%064x" % d == load("kats.json")["curv_digest_ext_doc_example"]
    # the compressed encoding would give a different digest
    d2 = hashlib.sha256(ec.compress(ec.G) + ec.compress(bp2) + b"\x0a").hexdigest()
    assert d2 != load("kats.json")["curv_digest_ext_doc_example"]


def test_wip_known_answer():
    """weighted_inner_product.rs:828-861 (the reference's only known-answer test)."""
    assert weighted_inner_product([3, 2, 1, 0], [1, 2, 3, 4], 2) == load("kats.json")["test_wip"] == 46


def test_result_scalar_is_try_and_increment():
    t = curv.sha256().chain(b"abc")
    want = int.from_bytes(hashlib.sha256(b"abc" + b"\0\0\0\0").digest(), "big")
    assert want < ec.Q and t.result_scalar() == want


def test_bigint_bytes():
    assert curv.bigint_to_bytes(0) == b"\x00"
    assert curv.bigint_to_bytes(0x0100) == b"\x01\x00"
    assert curv.bigint_from_bytes(b"\x00\x00\x05") == 5


def test_generate_random_point_and_duplicates():
    g_vec, h_vec, G, H = curv.bp_generators(8, 2)
    assert all(ec.on_curve(p) for p in g_vec + h_vec + [G, H])
    # h_vec label is n + i + seed (range_proof.rs:837-841): h_vec[i] == g_vec[i + n]
    assert h_vec[:8] == g_vec[8:16]
    assert G == ec.G
    # zero-padded short input and the doubling retry both terminate on a curve point
    assert ec.on_curve(curv.generate_random_point(b"\x01"))
    assert ec.on_curve(curv.generate_random_point(b"\x05" * 64))
    _, _, _, Hp = curv.bpplus_generators(8, 1)
    assert Hp == curv.label_point(curv.KZEN_SEED - 1)
