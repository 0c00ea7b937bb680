"""CPU oracle for the secp256k1 MSM hot path of ZenGo-X/bulletproofs.

TEST INFRASTRUCTURE ONLY.  Nothing under ``bulletproofs_amd/`` may import, link or execute
anything in this package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and there only as the checker / reported baseline.

What it is: a line-by-line Python restatement (Python ints) of
``/root/reference/src/proofs/{range_proof,inner_product,weighted_inner_product,range_proof_wip}.rs``
plus a restatement of what the external crate ``curv-kzen 0.10`` (Cargo.toml:22, NOT vendored
under /root/reference, Cargo.lock git-ignored) does at the boundary: secp256k1 group arithmetic,
``DigestExt`` transcript hashing and ``generate_random_point``.

Pinning status:
  * group arithmetic: pinned against OpenSSL libcrypto (``tests/test_oracle_ec.py``) and the
    standard secp256k1 constants;
  * ``weighted_inner_product``: pinned by the reference's only known-answer test
    (weighted_inner_product.rs:828-861, expected 46);
  * transcript byte format: partly pinned by the curv ``DigestExt`` doc-example digest
    (``tests/test_oracle_transcript.py``); the ``result_scalar`` try-and-increment suffix is a
    recollection of curv 0.10 -> **parity unpinned** for challenge bytes;
  * proof bytes: the reference's 38 other tests are prove->verify round trips under OS
    randomness, nothing pins bytes -> **parity unpinned**; accept/reject semantics are pinned by
    re-running those round trips (and tampered variants) through this restatement.
"""
