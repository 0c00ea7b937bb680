"""bulletproofs_amd — Python harness over libbpmsm.so (HIP kernels + C ABI, include/bpmsm.h).

The product is the shared library; this package only binds it with ctypes for the tests and
bench.py.  There is no CPU fallback: importing works anywhere (the library is loaded lazily), but
any compute call raises if libbpmsm.so is missing or no HIP device is present.
"""
from ._lib import (  # noqa: F401
    BpError,
    Context,
    Gens,
    generate_bases,
    msm_auto_window,
    noalloc_scratch_bytes,
    proof_to_serde,
    proof_from_serde,
    SERDE_JSON,
    SERDE_BINCODE,
    BP_EINVAL,
    KZEN_SEED,
    lib,
    lib_path,
    EXPORTED_SYMBOLS,
)
