"""Restatement of the curv-kzen 0.10 behaviours the reference relies on.

curv-kzen is an external crate (Cargo.toml:22: `curv-kzen = { version = "0.10",
default-features = false }`, feature `rust-gmp-kzen`), not vendored under /root/reference and not
buildable here (no cargo/rustc).  Everything here follows its published behaviour as recalled in
SURVEY.md appendix A and is anchored on the reference's call sites:

  * ``Sha256::new().chain_points([..]).chain_bigint(..).result_scalar()``
        range_proof.rs:124,127,171,214-218,591-593,656; inner_product.rs:117,192-194,282;
        range_proof_wip.rs:167,171,465-467,477; weighted_inner_product.rs:158,243,299-301,...
  * ``Sha512::new().chain_bigint(&label).result_bigint()`` + ``Converter::to_bytes``
        range_proof.rs:825-826
  * ``generate_random_point``  range_proof.rs:785-808

KAT that pins `chain_point`/`chain_bigint`/`result_bigint` byte formats: curv's DigestExt
doc-example digest (tests/test_oracle_transcript.py).  The 4-byte big-endian counter suffix of
``result_scalar`` (try-and-increment) is a recollection: parity unpinned.
"""
import hashlib

from . import ec

# result_scalar: h = H(state || i.to_be_bytes()) for i = 0u32.., first h[..32] that parses as a
# scalar (< q; zero allowed).  Isolated here so it can be corrected in one place.
RESULT_SCALAR_COUNTER_BYTES = 4


def bigint_to_bytes(n):
    """BigInt::to_bytes (GMP backend): minimal big-endian magnitude, zero -> b"\\x00"."""
    if n < 0:
        raise ValueError("negative BigInt")
    if n == 0:
        return b"\x00"
    return n.to_bytes((n.bit_length() + 7) // 8, "big")


def bigint_from_bytes(b):
    return int.from_bytes(bytes(b), "big")


class Transcript:
    """`Sha256::new()` / `Sha512::new()` with curv's DigestExt chaining."""

    def __init__(self, algo="sha256"):
        self._h = hashlib.new(algo)

    def chain(self, data):
        self._h.update(bytes(data))
        return self

    def chain_points(self, pts):
        for p in pts:
            self._h.update(ec.uncompressed(p))
        return self

    def chain_point(self, p):
        return self.chain_points([p])

    def chain_bigint(self, n):
        self._h.update(bigint_to_bytes(n))
        return self

    def result_bigint(self):
        return int.from_bytes(self._h.digest(), "big")

    def result_scalar(self):
        i = 0
        while True:
            h = self._h.copy()
            if RESULT_SCALAR_COUNTER_BYTES:
                h.update(i.to_bytes(RESULT_SCALAR_COUNTER_BYTES, "big"))
            v = int.from_bytes(h.digest()[:32], "big")
            if v < ec.Q:
                return v
            i += 1
            if not RESULT_SCALAR_COUNTER_BYTES:  # plain-reduction variant
                return v % ec.Q


def sha256():
    return Transcript("sha256")


def sha512():
    return Transcript("sha512")


def generate_random_point(data):
    """range_proof.rs:785-808 (identical to curv's secp256_k1::hash_to_curve::generate_random_point).

    Try to decompress 0x02 || data[0..32]; on failure data <- to_bytes(int(data) * 2 mod q), recurse.
    If data is shorter than 32 bytes the 33-byte buffer is zero-padded on the right.
    """
    data = bytes(data)
    while True:
        truncated = data[:32] if len(data) > 32 else data
        buf = bytearray(33)
        buf[0] = 2
        buf[1:1 + len(truncated)] = truncated
        try:
            return ec.decompress(bytes(buf))
        except ValueError:
            pass
        bn = bigint_from_bytes(data)
        data = bigint_to_bytes(bn * 2 % ec.Q)


def label_point(label):
    """`generate_random_point(&Converter::to_bytes(&Sha512::new().chain_bigint(&label).result_bigint()))`
    (range_proof.rs:825-826, 830-832): leading zero bytes of the digest are dropped by the
    BigInt round trip."""
    h = sha512().chain_bigint(label).result_bigint()
    return generate_random_point(bigint_to_bytes(h))


KZEN_SEED = int.from_bytes(bytes([75, 90, 101, 110]), "big")  # range_proof.rs:912-913


def bp_generators(n, m, seed=KZEN_SEED):
    """Input recipe of the reference's tests/benches for Bulletproofs
    (range_proof.rs:821-843, benches/range_proof.rs:43-67): returns (g_vec, h_vec, G, H).
    Note h_vec uses label n + i + seed (n = bit length), so h_vec[i] == g_vec[i+n] for i+n < nm."""
    nm = n * m
    G = ec.G
    H = label_point(1)
    g_vec = [label_point(i + seed) for i in range(nm)]
    h_vec = [label_point(n + i + seed) for i in range(nm)]
    return g_vec, h_vec, G, H


def bpplus_generators(n, m, seed=KZEN_SEED):
    """StatementRP::generate_bases (range_proof_wip.rs:48-91): H label = seed - 1 mod q."""
    nm = n * m
    G = ec.G
    H = label_point((seed - 1) % ec.Q)
    g_vec = [label_point(i + seed) for i in range(nm)]
    h_vec = [label_point(n + i + seed) for i in range(nm)]
    return g_vec, h_vec, G, H
