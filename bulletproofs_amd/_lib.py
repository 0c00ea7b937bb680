"""ctypes binding of include/bpmsm.h."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

BP_OK = 0
BP_EINVAL, BP_EPOINT, BP_EHIP, BP_ERCCL, BP_ENOMEM = -1, -2, -3, -4, -5

# every symbol include/bpmsm.h declares (tests/test_abi.py checks the header against this list
# and the loaded library against both)
EXPORTED_SYMBOLS = [
    "bp_ctx_create",
    "bp_ctx_destroy",
    "bp_last_error",
    "bp_device_count",
    "bp_msm",
    "bp_msm_device",
    "bp_msm_set_window",
    "bp_msm_set_path",
    "bp_msm_set_pipeline",
    "bp_proto_set_path",
    "bp_msm_batch",
    "bp_msm_auto_window",
    "bp_msm_partial_bytes",
    "bp_msm_partial_device",
    "bp_msm_partial",
    "bp_msm_combine",
    "bp_msm_partial_device_out",
    "bp_msm_combine_device",
    "bp_ctx_record_event",
    "bp_ctx_wait_event",
    "bp_gens_create",
    "bp_gens_destroy",
    "bp_gens_table_bytes",
    "bp_rangeproof_verify_batch",
    "bp_rangeproof_verify_batch_device",
    "bp_rangeproof_verify_batch_rlc",
    "bp_wip_rangeproof_verify_batch",
    "bp_wip_rangeproof_verify_batch_device",
    "bp_wip_rangeproof_verify_batch_rlc",
    "bp_wip_rangeproof_verify_batch_rlc_device",
    "bp_rangeproof_verify_batch_rlc_device",
    "bp_ctx_sync",
    "bp_rangeproof_verify_batch_device_noalloc",
    "bp_noalloc_scratch_bytes",
    "bp_verify_set_lanes",
    "bp_verify_set_overlap",
    "bp_scalar_mul_batch",
    "bp_fold_bases",
    "bp_fixed_base_mul_batch",
    "bp_generate_bases",
    "bp_generate_bases_device",
    "bp_rangeproof_prove",
    "bp_rangeproof_prove_batch",
    "bp_wip_rangeproof_prove_batch",
    "bp_rangeproof_verify",
    "bp_wip_rangeproof_prove",
    "bp_wip_rangeproof_verify",
    "bp_rangeproof_verify_batch_mixed",
    "bp_wip_rangeproof_verify_batch_mixed",
    "bp_ipa_prove",
    "bp_ipa_verify",
    "bp_wipa_prove",
    "bp_wipa_verify",
    "bp_proof_to_serde",
    "bp_proof_from_serde",
    "bp_profile_enable",
    "bp_profile_reset",
    "bp_profile_get",
]


class BpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("bpmsm error %d: %s" % (code, msg))
        self.code = code


def lib_path():
    # BPMSM_LIB selects another build of the same library (scripts/opcount.py loads the instrumented libbpmsm_opcount.so)
    return os.environ.get("BPMSM_LIB") or os.path.join(_HERE, "libbpmsm.so")


_lib = None


def lib():
    """Load libbpmsm.so (built in-tree by `make -C bulletproofs_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise BpError(BP_EHIP, "libbpmsm.so not built (%s); run __graft_entry__.build() — there is no CPU fallback" % path)
    L = ctypes.CDLL(path)
    vp, cp, sz, i32 = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int
    L.bp_ctx_create.argtypes = [i32, ctypes.POINTER(vp)]
    L.bp_ctx_create.restype = i32
    L.bp_ctx_destroy.argtypes = [vp]
    L.bp_ctx_destroy.restype = None
    L.bp_last_error.argtypes = []
    L.bp_last_error.restype = cp
    L.bp_device_count.argtypes = []
    L.bp_device_count.restype = i32
    L.bp_msm.argtypes = [vp, cp, cp, sz, cp, cp]
    L.bp_msm.restype = i32
    L.bp_msm_device.argtypes = [vp, vp, vp, sz, cp, cp]
    L.bp_msm_device.restype = i32
    L.bp_msm_set_window.argtypes = [vp, i32]
    L.bp_msm_set_window.restype = None
    L.bp_msm_set_path.argtypes = [vp, i32]
    L.bp_msm_set_path.restype = None
    L.bp_msm_set_pipeline.argtypes = [vp, i32]
    L.bp_msm_set_pipeline.restype = None
    L.bp_proto_set_path.argtypes = [vp, i32]
    L.bp_proto_set_path.restype = None
    L.bp_gens_create.argtypes = [vp, cp, cp, sz, cp, cp, i32, ctypes.POINTER(vp)]
    L.bp_gens_create.restype = i32
    L.bp_gens_destroy.argtypes = [vp]
    L.bp_gens_destroy.restype = None
    L.bp_gens_table_bytes.argtypes = [vp]
    L.bp_gens_table_bytes.restype = sz
    L.bp_rangeproof_verify_batch.argtypes = [vp, vp, i32, i32, cp, sz, cp, sz, cp]
    L.bp_rangeproof_verify_batch.restype = i32
    L.bp_rangeproof_verify_batch_device.argtypes = [vp, vp, i32, i32, vp, sz, vp, sz, vp]
    L.bp_rangeproof_verify_batch_device.restype = i32
    L.bp_wip_rangeproof_verify_batch.argtypes = [vp, vp, i32, i32, cp, sz, cp, sz, cp]
    L.bp_wip_rangeproof_verify_batch.restype = i32
    L.bp_wip_rangeproof_verify_batch_device.argtypes = [vp, vp, i32, i32, vp, sz, vp, sz, vp]
    L.bp_wip_rangeproof_verify_batch_device.restype = i32
    L.bp_wip_rangeproof_verify_batch_rlc.argtypes = [vp, vp, i32, i32, cp, sz, cp, sz, cp, cp, ctypes.POINTER(i32)]
    L.bp_wip_rangeproof_verify_batch_rlc.restype = i32
    L.bp_wip_rangeproof_verify_batch_rlc_device.argtypes = [vp, vp, i32, i32, vp, sz, vp, sz, cp, vp, ctypes.POINTER(i32)]
    L.bp_wip_rangeproof_verify_batch_rlc_device.restype = i32
    L.bp_rangeproof_verify_batch_rlc.argtypes = [vp, vp, i32, i32, cp, sz, cp, sz, cp, cp, ctypes.POINTER(i32)]
    L.bp_rangeproof_verify_batch_rlc.restype = i32
    L.bp_rangeproof_verify_batch_rlc_device.argtypes = [vp, vp, i32, i32, vp, sz, vp, sz, cp, vp, ctypes.POINTER(i32)]
    L.bp_rangeproof_verify_batch_rlc_device.restype = i32
    L.bp_rangeproof_verify_batch_device_noalloc.argtypes = [vp, vp, i32, i32, vp, sz, vp, sz, vp]
    L.bp_rangeproof_verify_batch_device_noalloc.restype = i32
    L.bp_noalloc_scratch_bytes.argtypes = []
    L.bp_noalloc_scratch_bytes.restype = sz
    L.bp_ctx_sync.argtypes = [vp]
    L.bp_ctx_sync.restype = i32
    L.bp_verify_set_lanes.argtypes = [vp, i32]
    L.bp_verify_set_lanes.restype = None
    L.bp_verify_set_overlap.argtypes = [vp, i32]
    L.bp_verify_set_overlap.restype = None
    L.bp_scalar_mul_batch.argtypes = [vp, cp, cp, sz, cp]
    L.bp_scalar_mul_batch.restype = i32
    L.bp_fold_bases.argtypes = [vp, cp, cp, sz, cp, cp, cp]
    L.bp_fold_bases.restype = i32
    L.bp_fixed_base_mul_batch.argtypes = [vp, cp, sz, cp]
    L.bp_fixed_base_mul_batch.restype = i32
    L.bp_msm_auto_window.argtypes = [sz]
    L.bp_msm_auto_window.restype = i32
    L.bp_msm_partial_bytes.argtypes = [i32]
    L.bp_msm_partial_bytes.restype = sz
    L.bp_msm_partial_device.argtypes = [vp, vp, vp, sz, i32, cp]
    L.bp_msm_partial_device.restype = i32
    L.bp_msm_partial.argtypes = [vp, cp, cp, sz, i32, cp]
    L.bp_msm_partial.restype = i32
    L.bp_msm_combine.argtypes = [vp, cp, sz, i32, cp, cp]
    L.bp_msm_combine.restype = i32
    L.bp_msm_partial_device_out.argtypes = [vp, vp, vp, sz, i32, vp]
    L.bp_msm_partial_device_out.restype = i32
    L.bp_msm_combine_device.argtypes = [vp, vp, sz, i32, cp, cp]
    L.bp_msm_combine_device.restype = i32
    L.bp_ctx_record_event.argtypes = [vp, vp]
    L.bp_ctx_record_event.restype = i32
    L.bp_ctx_wait_event.argtypes = [vp, vp]
    L.bp_ctx_wait_event.restype = i32
    L.bp_msm_batch.argtypes = [vp, vp, vp, sz, cp]
    L.bp_msm_batch.restype = i32
    L.bp_generate_bases.argtypes = [i32, ctypes.c_uint64, sz, sz, cp, cp, cp, cp]
    L.bp_generate_bases.restype = i32
    for f in (L.bp_rangeproof_prove_batch, L.bp_wip_rangeproof_prove_batch):
        f.argtypes = [vp, vp, i32, i32, sz, cp, cp, cp, sz, cp, sz, ctypes.POINTER(i32)]
        f.restype = i32
    L.bp_generate_bases_device.argtypes = [vp, i32, ctypes.c_uint64, sz, sz, cp, cp, cp, cp]
    L.bp_generate_bases_device.restype = i32
    prove_args = [vp, vp, i32, cp, cp, i32, cp, sz, cp, sz, ctypes.POINTER(sz)]
    verify_args = [vp, vp, i32, i32, cp, sz, cp, i32, cp]
    L.bp_rangeproof_prove.argtypes = prove_args
    L.bp_rangeproof_prove.restype = i32
    L.bp_wip_rangeproof_prove.argtypes = prove_args
    L.bp_wip_rangeproof_prove.restype = i32
    L.bp_rangeproof_verify.argtypes = verify_args
    L.bp_rangeproof_verify.restype = i32
    L.bp_wip_rangeproof_verify.argtypes = verify_args
    L.bp_wip_rangeproof_verify.restype = i32
    for f in (L.bp_rangeproof_verify_batch_mixed, L.bp_wip_rangeproof_verify_batch_mixed):
        f.argtypes = [vp, vp, i32, cp, ctypes.POINTER(sz), cp, ctypes.POINTER(ctypes.c_uint32), sz, cp]
        f.restype = i32
    L.bp_ipa_prove.argtypes = [vp, cp, cp, sz, cp, cp, cp, cp, cp, sz, ctypes.POINTER(sz)]
    L.bp_ipa_prove.restype = i32
    L.bp_ipa_verify.argtypes = [vp, cp, cp, sz, cp, cp, cp, sz, i32, cp]
    L.bp_ipa_verify.restype = i32
    L.bp_wipa_prove.argtypes = [vp, cp, cp, sz, cp, cp, cp, cp, cp, cp, cp, cp, sz, cp, sz, ctypes.POINTER(sz)]
    L.bp_wipa_prove.restype = i32
    L.bp_wipa_verify.argtypes = [vp, cp, cp, sz, cp, cp, cp, cp, cp, sz, i32, cp]
    L.bp_wipa_verify.restype = i32
    L.bp_proof_to_serde.argtypes = [i32, i32, cp, sz, cp, sz, ctypes.POINTER(sz)]
    L.bp_proof_to_serde.restype = i32
    L.bp_proof_from_serde.argtypes = [i32, i32, cp, sz, cp, sz, ctypes.POINTER(sz)]
    L.bp_proof_from_serde.restype = i32
    L.bp_profile_enable.argtypes = [vp, i32]
    L.bp_profile_enable.restype = None
    L.bp_profile_reset.argtypes = [vp]
    L.bp_profile_reset.restype = None
    L.bp_profile_get.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long)]
    L.bp_profile_get.restype = i32
    _lib = L
    return L


def _check(rc):
    if rc != BP_OK:
        raise BpError(rc, lib().bp_last_error().decode(errors="replace"))


class MsmJob(ctypes.Structure):
    """`bp_msm_job` of include/bpmsm.h."""
    _fields_ = [("gen_scalars_be", ctypes.c_void_p), ("extra_points_xy", ctypes.c_void_p), ("extra_scalars_be", ctypes.c_void_p),
                ("n_extra", ctypes.c_size_t)]


class Context:
    """`bp_ctx`: one HIP stream + workspaces on one device."""

    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        _check(lib().bp_ctx_create(device, ctypes.byref(self._h)))
        self.device = device

    def close(self):
        if self._h:
            lib().bp_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_window(self, c):
        lib().bp_msm_set_window(self._h, c)

    def set_path(self, path):
        """0 = automatic, 1 = LDS slice path, 2 = global sorted path (testing / tuning)."""
        lib().bp_msm_set_path(self._h, path)

    def set_msm_pipeline(self, mode):
        """0 = window groups on side streams (lowest latency of one large MSM), 1 = one group (several contexts in flight)."""
        lib().bp_msm_set_pipeline(self._h, mode)

    def set_proto_path(self, path):
        """0 = generator MSMs of prove/verify from the fixed-base table (default), 1 = reference-structured rounds."""
        lib().bp_proto_set_path(self._h, path)

    def msm(self, points_xy, scalars_be, init_xy=None):
        """out = init + sum scalars[i]*points[i]; bytes in, 64 bytes out (see include/bpmsm.h)."""
        points_xy, scalars_be = bytes(points_xy), bytes(scalars_be)
        n = len(scalars_be) // 32
        if len(points_xy) != 64 * n or len(scalars_be) != 32 * n:
            raise BpError(BP_EINVAL, "points/scalars length mismatch")
        out = ctypes.create_string_buffer(64)
        _check(lib().bp_msm(self._h, points_xy, scalars_be, n, init_xy, out))
        return out.raw

    def msm_device(self, d_points, d_scalars, n, init_xy=None):
        """Same with device pointers (ints), e.g. torch tensors' data_ptr()."""
        out = ctypes.create_string_buffer(64)
        _check(lib().bp_msm_device(self._h, ctypes.c_void_p(d_points), ctypes.c_void_p(d_scalars), n, init_xy, out))
        return out.raw

    def sync(self):
        _check(lib().bp_ctx_sync(self._h))

    def profile_enable(self, on=True):
        lib().bp_profile_enable(self._h, 1 if on else 0)

    def profile_reset(self):
        lib().bp_profile_reset(self._h)

    def profile_get(self, kernel_id):
        """(total_ms, launches) accumulated for one kernel id (include/bpmsm.h BP_PROF_*)."""
        ms, n = ctypes.c_double(), ctypes.c_long()
        _check(lib().bp_profile_get(self._h, kernel_id, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def set_verify_lanes(self, lanes):
        lib().bp_verify_set_lanes(self._h, lanes)

    def set_verify_overlap(self, on):
        """Scalar stage of successive device batch verifications on a side stream (default on; bp_verify_set_overlap)."""
        lib().bp_verify_set_overlap(self._h, 1 if on else 0)

    def rangeproof_verify_batch(self, gens, n_bits, m, proofs, coms_xy):
        """`RangeProof::aggregated_verify` for len(proofs)//proof_len proofs -> list of verdict codes."""
        proofs, coms_xy = bytes(proofs), bytes(coms_xy)
        nm = n_bits * m
        k = max(nm.bit_length() - 1, 0)
        proof_len = 292 + 66 * k
        nproofs = len(proofs) // proof_len if proof_len else 0
        if nproofs * proof_len != len(proofs) or len(coms_xy) != nproofs * m * 64:
            raise BpError(BP_EINVAL, "proofs/commitments length mismatch")
        out = ctypes.create_string_buffer(max(nproofs, 1))
        _check(lib().bp_rangeproof_verify_batch(self._h, gens._h, n_bits, m, proofs, proof_len, coms_xy, nproofs, out))
        return list(out.raw[:nproofs])

    def wip_rangeproof_verify_batch(self, gens, n_bits, m, proofs, coms_xy):
        """`RangeProofWIP::aggregated_verify` (Bulletproofs+) for len(proofs)//proof_len proofs -> list of verdict codes."""
        proofs, coms_xy = bytes(proofs), bytes(coms_xy)
        nm = n_bits * m
        k = max(nm.bit_length() - 1, 0)
        proof_len = 195 + 66 * k
        nproofs = len(proofs) // proof_len if proof_len else 0
        if nproofs * proof_len != len(proofs) or len(coms_xy) != nproofs * m * 64:
            raise BpError(BP_EINVAL, "proofs/commitments length mismatch")
        out = ctypes.create_string_buffer(max(nproofs, 1))
        _check(lib().bp_wip_rangeproof_verify_batch(self._h, gens._h, n_bits, m, proofs, proof_len, coms_xy, nproofs, out))
        return list(out.raw[:nproofs])

    def wip_rangeproof_verify_batch_device(self, gens, n_bits, m, d_proofs, proof_len, d_coms, nproofs, d_verdicts):
        """Asynchronous device-pointer variant (ints); call sync() before reading verdicts."""
        _check(lib().bp_wip_rangeproof_verify_batch_device(self._h, gens._h, n_bits, m, ctypes.c_void_p(d_proofs), proof_len,
                                                           ctypes.c_void_p(d_coms), nproofs, ctypes.c_void_p(d_verdicts)))

    def wip_rangeproof_verify_batch_rlc(self, gens, n_bits, m, proofs, coms_xy, seed):
        """Bulletproofs+ verdicts through one random-linear-combination MSM -> (list of verdict codes, used_fallback)."""
        proofs, coms_xy, seed = bytes(proofs), bytes(coms_xy), bytes(seed)
        nm = n_bits * m
        k = max(nm.bit_length() - 1, 0)
        proof_len = 195 + 66 * k
        nproofs = len(proofs) // proof_len if proof_len else 0
        if nproofs * proof_len != len(proofs) or len(coms_xy) != nproofs * m * 64 or len(seed) != 32:
            raise BpError(BP_EINVAL, "proofs/commitments/seed length mismatch")
        out = ctypes.create_string_buffer(max(nproofs, 1))
        fb = ctypes.c_int32(0)
        _check(lib().bp_wip_rangeproof_verify_batch_rlc(self._h, gens._h, n_bits, m, proofs, proof_len, coms_xy, nproofs, seed, out,
                                                        ctypes.byref(fb)))
        return list(out.raw[:nproofs]), bool(fb.value)

    def wip_rangeproof_verify_batch_rlc_device(self, gens, n_bits, m, d_proofs, proof_len, d_coms, nproofs, seed, d_verdicts):
        fb = ctypes.c_int32(0)
        _check(lib().bp_wip_rangeproof_verify_batch_rlc_device(self._h, gens._h, n_bits, m, d_proofs, proof_len, d_coms, nproofs, bytes(seed),
                                                               d_verdicts, ctypes.byref(fb)))
        return bool(fb.value)

    def rangeproof_verify_batch_rlc(self, gens, n_bits, m, proofs, coms_xy, seed):
        """Same verdicts through one random-linear-combination MSM (per-proof fallback when it fails)
        -> (list of verdict codes, used_fallback)."""
        proofs, coms_xy, seed = bytes(proofs), bytes(coms_xy), bytes(seed)
        nm = n_bits * m
        k = max(nm.bit_length() - 1, 0)
        proof_len = 292 + 66 * k
        nproofs = len(proofs) // proof_len if proof_len else 0
        if nproofs * proof_len != len(proofs) or len(coms_xy) != nproofs * m * 64 or len(seed) != 32:
            raise BpError(BP_EINVAL, "proofs/commitments/seed length mismatch")
        out = ctypes.create_string_buffer(max(nproofs, 1))
        fb = ctypes.c_int32(0)
        _check(lib().bp_rangeproof_verify_batch_rlc(self._h, gens._h, n_bits, m, proofs, proof_len, coms_xy, nproofs, seed, out, ctypes.byref(fb)))
        return list(out.raw[:nproofs]), bool(fb.value)

    def rangeproof_verify_batch_rlc_device(self, gens, n_bits, m, d_proofs, proof_len, d_coms, nproofs, seed, d_verdicts):
        fb = ctypes.c_int32(0)
        _check(lib().bp_rangeproof_verify_batch_rlc_device(self._h, gens._h, n_bits, m, d_proofs, proof_len, d_coms, nproofs, bytes(seed),
                                                           d_verdicts, ctypes.byref(fb)))
        return bool(fb.value)

    def scalar_mul_batch(self, points_xy, scalars_be):
        n = len(scalars_be) // 32
        out = ctypes.create_string_buffer(max(64 * n, 1))
        _check(lib().bp_scalar_mul_batch(self._h, bytes(points_xy), bytes(scalars_be), n, out))
        return out.raw[:64 * n]

    def fold_bases(self, lo_xy, hi_xy, s_lo, s_hi):
        n = len(lo_xy) // 64
        out = ctypes.create_string_buffer(max(64 * n, 1))
        _check(lib().bp_fold_bases(self._h, bytes(lo_xy), bytes(hi_xy), n, bytes(s_lo), bytes(s_hi), out))
        return out.raw[:64 * n]

    def msm_partial_device(self, d_points, d_scalars, n, c):
        """Rank-local window sums of a sharded MSM (device pointers in, opaque host bytes out)."""
        out = ctypes.create_string_buffer(lib().bp_msm_partial_bytes(c))
        _check(lib().bp_msm_partial_device(self._h, d_points, d_scalars, n, c, out))
        return out.raw

    def msm_partial_device_out(self, d_points, d_scalars, n, c, d_partial_out):
        """bp_msm_partial_device_out: the partial stays in device memory (d_partial_out: int device pointer); asynchronous."""
        _check(lib().bp_msm_partial_device_out(self._h, ctypes.c_void_p(d_points), ctypes.c_void_p(d_scalars), n, c, ctypes.c_void_p(d_partial_out)))

    def msm_combine_device(self, d_partials, nparts, c, init_xy=None):
        """bp_msm_combine_device: partials of all ranks in device memory -> the 64-byte result."""
        out = ctypes.create_string_buffer(64)
        _check(lib().bp_msm_combine_device(self._h, ctypes.c_void_p(d_partials), nparts, c, init_xy, out))
        return out.raw

    def record_event(self, hip_event):
        """Record the caller's hipEvent_t (int handle, e.g. torch.cuda.Event.cuda_event) on the context's stream."""
        _check(lib().bp_ctx_record_event(self._h, ctypes.c_void_p(hip_event)))

    def wait_event(self, hip_event):
        """Order the context's later work (its stream and the verifier's side lane) after the caller's hipEvent_t."""
        _check(lib().bp_ctx_wait_event(self._h, ctypes.c_void_p(hip_event)))

    def msm_partial(self, points_xy, scalars_be, c):
        n = len(scalars_be) // 32
        out = ctypes.create_string_buffer(lib().bp_msm_partial_bytes(c))
        _check(lib().bp_msm_partial(self._h, bytes(points_xy), bytes(scalars_be), n, c, out))
        return out.raw

    def msm_combine(self, partials, nparts, c, init_xy=None):
        out = ctypes.create_string_buffer(64)
        _check(lib().bp_msm_combine(self._h, bytes(partials), nparts, c, init_xy, out))
        return out.raw

    def msm_batch(self, gens, jobs):
        """jobs: list of (gen_scalars_be or None, extra_points_xy, extra_scalars_be); returns 64 bytes per job."""
        n = len(jobs)
        arr = (MsmJob * max(n, 1))()
        keep = []
        for i, (gs, xp, xs) in enumerate(jobs):
            xp, xs = bytes(xp or b""), bytes(xs or b"")
            bufs = [ctypes.create_string_buffer(bytes(gs), len(gs)) if gs is not None else None,
                    ctypes.create_string_buffer(xp, max(len(xp), 1)), ctypes.create_string_buffer(xs, max(len(xs), 1))]
            keep.append(bufs)
            arr[i].gen_scalars_be = ctypes.cast(bufs[0], ctypes.c_void_p) if bufs[0] is not None else None
            arr[i].extra_points_xy = ctypes.cast(bufs[1], ctypes.c_void_p)
            arr[i].extra_scalars_be = ctypes.cast(bufs[2], ctypes.c_void_p)
            arr[i].n_extra = len(xp) // 64
        out = ctypes.create_string_buffer(max(64 * n, 1))
        _check(lib().bp_msm_batch(self._h, gens._h, ctypes.byref(arr), n, out))
        return out.raw[:64 * n]

    def fixed_base_mul_batch(self, scalars_be):
        n = len(scalars_be) // 32
        out = ctypes.create_string_buffer(max(64 * n, 1))
        _check(lib().bp_fixed_base_mul_batch(self._h, bytes(scalars_be), n, out))
        return out.raw[:64 * n]

    def generate_bases(self, kind, seed, n_bits, m):
        """`generate_bases` computed on the GPU (bp_generate_bases_device): same bytes as the host recipe."""
        nm = n_bits * m
        g = ctypes.create_string_buffer(64 * nm)
        h = ctypes.create_string_buffer(64 * nm)
        G = ctypes.create_string_buffer(64)
        H = ctypes.create_string_buffer(64)
        _check(lib().bp_generate_bases_device(self._h, kind, seed, n_bits, m, g, h, G, H))
        return g.raw, h.raw, G.raw, H.raw

    def _prove(self, fn, gens, n_bits, secrets_be, blindings_be, tape):
        m = len(secrets_be) // 32
        out = ctypes.create_string_buffer(4096)
        ln = ctypes.c_size_t(0)
        _check(fn(self._h, gens._h, n_bits, bytes(secrets_be), bytes(blindings_be), m, bytes(tape), len(tape), out, 4096, ctypes.byref(ln)))
        return out.raw[:ln.value]

    def rangeproof_prove(self, gens, n_bits, secrets_be, blindings_be, tape):
        """`RangeProof::prove`; tape = alpha, rho, SR[nm], SL[nm], tau1, tau2 (32-byte scalars)."""
        return self._prove(lib().bp_rangeproof_prove, gens, n_bits, secrets_be, blindings_be, tape)

    def wip_rangeproof_prove(self, gens, n_bits, secrets_be, blindings_be, tape):
        """`RangeProofWIP::prove`; tape = alpha, (d_L, d_R) per round, r, s, delta, eta."""
        return self._prove(lib().bp_wip_rangeproof_prove, gens, n_bits, secrets_be, blindings_be, tape)

    def _prove_batch(self, fn, plen0, gens, n_bits, m, secrets_be, blindings_be, tapes, tape_len):
        nproofs = len(secrets_be) // (32 * m)
        nm = n_bits * m
        proof_len = plen0 + 66 * max(nm.bit_length() - 1, 0)
        if len(secrets_be) != nproofs * m * 32 or len(blindings_be) != nproofs * m * 32 or len(tapes) != nproofs * tape_len:
            raise BpError(BP_EINVAL, "secrets/blindings/tapes length mismatch")
        out = ctypes.create_string_buffer(max(nproofs * proof_len, 1))
        st = (ctypes.c_int32 * max(nproofs, 1))()
        _check(fn(self._h, gens._h, n_bits, m, nproofs, bytes(secrets_be), bytes(blindings_be), bytes(tapes), tape_len, out, proof_len, st))
        raw = out.raw  # one copy: .raw copies the whole buffer on every access
        return [raw[i * proof_len:(i + 1) * proof_len] for i in range(nproofs)]

    def rangeproof_prove_batch(self, gens, n_bits, m, secrets_be, blindings_be, tapes, tape_len):
        """`RangeProof::prove` for len(secrets_be)//(32 m) proofs (one GPU launch per protocol step for all) -> list of proofs."""
        return self._prove_batch(lib().bp_rangeproof_prove_batch, 292, gens, n_bits, m, secrets_be, blindings_be, tapes, tape_len)

    def wip_rangeproof_prove_batch(self, gens, n_bits, m, secrets_be, blindings_be, tapes, tape_len):
        """`RangeProofWIP::prove` for a batch -> list of proofs."""
        return self._prove_batch(lib().bp_wip_rangeproof_prove_batch, 195, gens, n_bits, m, secrets_be, blindings_be, tapes, tape_len)

    def rangeproof_verify(self, gens, n_bits, m, proof, coms_xy, mode=2):
        """mode 0 = verify, 1 = fast_verify, 2 = aggregated_verify -> verdict code."""
        v = ctypes.create_string_buffer(1)
        _check(lib().bp_rangeproof_verify(self._h, gens._h, n_bits, m, bytes(proof), len(proof), bytes(coms_xy), mode, v))
        return v.raw[0]

    def wip_rangeproof_verify(self, gens, n_bits, m, proof, coms_xy, mode=2):
        """mode 0 = verify, 2 = aggregated_verify -> verdict code."""
        v = ctypes.create_string_buffer(1)
        _check(lib().bp_wip_rangeproof_verify(self._h, gens._h, n_bits, m, bytes(proof), len(proof), bytes(coms_xy), mode, v))
        return v.raw[0]

    def rangeproof_verify_batch_mixed(self, gens, n_bits, proofs, ms, coms_xy, wip=False):
        """One call for proofs with mixed aggregation sizes: proofs = list of proof bytes, ms = their m values,
        coms_xy = all commitments concatenated in proof order -> list of verdict codes."""
        n = len(proofs)
        off = (ctypes.c_size_t * (n + 1))()
        for i, p in enumerate(proofs):
            off[i + 1] = off[i] + len(p)
        marr = (ctypes.c_uint32 * max(n, 1))(*ms)
        out = ctypes.create_string_buffer(max(n, 1))
        fn = lib().bp_wip_rangeproof_verify_batch_mixed if wip else lib().bp_rangeproof_verify_batch_mixed
        _check(fn(self._h, gens._h, n_bits, b"".join(proofs), off, bytes(coms_xy), marr, n, out))
        return list(out.raw[:n])

    def ipa_prove(self, g_xy, h_xy, ux_xy, a_be, b_be, P_xy=None):
        """`InnerProductArg::prove` over caller-supplied bases -> L[k] | R[k] | a | b bytes."""
        n = len(a_be) // 32
        out = ctypes.create_string_buffer(66 * 32 + 64)
        ln = ctypes.c_size_t(0)
        _check(lib().bp_ipa_prove(self._h, bytes(g_xy), bytes(h_xy), n, bytes(ux_xy), P_xy, bytes(a_be), bytes(b_be), out, len(out), ctypes.byref(ln)))
        return out.raw[:ln.value]

    def ipa_verify(self, g_xy, hi_tag_xy, ux_xy, P_xy, arg, mode=0):
        """mode 0 = `InnerProductArg::verify` (recursive), 1 = `fast_verify` -> verdict code (0 / 1 / 255)."""
        v = ctypes.create_string_buffer(1)
        _check(lib().bp_ipa_verify(self._h, bytes(g_xy), bytes(hi_tag_xy), len(g_xy) // 64, bytes(ux_xy), bytes(P_xy), bytes(arg), len(arg), mode, v))
        return v.raw[0]

    def wipa_prove(self, g_xy, h_xy, g_pt, h_pt, a_be, b_be, alpha_be, y_be, tape, P_xy=None):
        """`WeightedInnerProdArg::prove` over caller-supplied bases; tape = (d_L, d_R) per round, r, s, delta, eta."""
        n = len(a_be) // 32
        out = ctypes.create_string_buffer(66 * 32 + 162)
        ln = ctypes.c_size_t(0)
        _check(lib().bp_wipa_prove(self._h, bytes(g_xy), bytes(h_xy), n, bytes(g_pt), bytes(h_pt), P_xy, bytes(a_be), bytes(b_be), bytes(alpha_be),
                                   bytes(y_be), bytes(tape), len(tape), out, len(out), ctypes.byref(ln)))
        return out.raw[:ln.value]

    def wipa_verify(self, g_xy, hi_tag_xy, g_pt, h_pt, P_xy, y_be, arg, mode=0):
        """mode 0 = `WeightedInnerProdArg::verify` (recursive), 1 = `fast_verify` -> verdict code (0 / 2 / 255)."""
        v = ctypes.create_string_buffer(1)
        _check(lib().bp_wipa_verify(self._h, bytes(g_xy), bytes(hi_tag_xy), len(g_xy) // 64, bytes(g_pt), bytes(h_pt), bytes(P_xy), bytes(y_be),
                                    bytes(arg), len(arg), mode, v))
        return v.raw[0]

    def rangeproof_verify_batch_device(self, gens, n_bits, m, d_proofs, proof_len, d_coms, nproofs, d_verdicts):
        """Asynchronous device-pointer variant (ints); call sync() before reading verdicts."""
        _check(lib().bp_rangeproof_verify_batch_device(self._h, gens._h, n_bits, m, ctypes.c_void_p(d_proofs), proof_len,
                                                       ctypes.c_void_p(d_coms), nproofs, ctypes.c_void_p(d_verdicts)))

    def rangeproof_verify_batch_device_noalloc(self, gens, n_bits, m, d_proofs, proof_len, d_coms, nproofs, d_verdicts):
        """rangeproof_verify_batch_device with its workspaces in the library's fixed device array (no hipMalloc)."""
        _check(lib().bp_rangeproof_verify_batch_device_noalloc(self._h, gens._h, n_bits, m, ctypes.c_void_p(d_proofs), proof_len,
                                                               ctypes.c_void_p(d_coms), nproofs, ctypes.c_void_p(d_verdicts)))


def noalloc_scratch_bytes():
    return lib().bp_noalloc_scratch_bytes()


SERDE_JSON, SERDE_BINCODE = 0, 1


def proof_to_serde(kind, fmt, proof):
    """Canonical proof bytes -> the serde encoding of RangeProof (kind 0) / RangeProofWIP (kind 1); host-only, needs no GPU."""
    need = ctypes.c_size_t(0)
    proof = bytes(proof)
    rc = lib().bp_proof_to_serde(kind, fmt, proof, len(proof), None, 0, ctypes.byref(need))
    if need.value == 0:
        _check(rc)
    out = ctypes.create_string_buffer(need.value)
    _check(lib().bp_proof_to_serde(kind, fmt, proof, len(proof), out, need.value, ctypes.byref(need)))
    return out.raw[:need.value]


def proof_from_serde(kind, fmt, data):
    """serde encoding (JSON text / bincode bytes) -> canonical proof bytes; host-only, needs no GPU."""
    need = ctypes.c_size_t(0)
    data = bytes(data)
    rc = lib().bp_proof_from_serde(kind, fmt, data, len(data), None, 0, ctypes.byref(need))
    if need.value == 0:
        _check(rc)
    out = ctypes.create_string_buffer(need.value)
    _check(lib().bp_proof_from_serde(kind, fmt, data, len(data), out, need.value, ctypes.byref(need)))
    return out.raw[:need.value]


def msm_auto_window(n):
    """Window width the planner picks for an n-term MSM (same on every rank for the same n)."""
    return lib().bp_msm_auto_window(n)


def generate_bases(kind, seed, n_bits, m):
    """(g_xy, h_xy, G_xy, H_xy) bytes of the reference's generator recipe; kind 0 = Bulletproofs
    tests/benches, 1 = StatementRP::generate_bases (Bulletproofs+).  Host-only, needs no GPU."""
    nm = n_bits * m
    g = ctypes.create_string_buffer(64 * nm)
    h = ctypes.create_string_buffer(64 * nm)
    G = ctypes.create_string_buffer(64)
    H = ctypes.create_string_buffer(64)
    _check(lib().bp_generate_bases(kind, seed, n_bits, m, g, h, G, H))
    return g.raw, h.raw, G.raw, H.raw


KZEN_SEED = int.from_bytes(bytes([75, 90, 101, 110]), "big")  # range_proof.rs:912-913


class Gens:
    """`bp_gens`: g_vec, h_vec, G, H uploaded once and expanded into the fixed-base table."""

    def __init__(self, ctx, g_xy, h_xy, G_xy, H_xy, window_bits=0):
        g_xy, h_xy = bytes(g_xy), bytes(h_xy)
        if len(g_xy) != len(h_xy) or len(g_xy) % 64:
            raise BpError(BP_EINVAL, "g_vec/h_vec length mismatch")
        self._h = ctypes.c_void_p()
        self.nm = len(g_xy) // 64
        _check(lib().bp_gens_create(ctx._h, g_xy, h_xy, self.nm, bytes(G_xy), bytes(H_xy), window_bits, ctypes.byref(self._h)))

    @property
    def table_bytes(self):
        return lib().bp_gens_table_bytes(self._h)

    def close(self):
        if self._h:
            lib().bp_gens_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
