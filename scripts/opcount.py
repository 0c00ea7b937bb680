#!/usr/bin/env python3
"""Executed-work count for bench.py's `frac_executed`: runs the verifier's MSM kernel and the raw MSM pipeline in the
INSTRUMENTED build (make -C bulletproofs_amd/csrc opcount -> libbpmsm_opcount.so, fp.h BP_OPCOUNT: every lane counts its
fp_mul / fp_sqr calls) and converts to v_mad_u64_u32 lane-results with the static per-call counts of the ISA
(fp_mul = 102, fp_sqr = 65 v_mad_u64_u32 per call incl. the reduction; profiles/r02_fp_isa.txt).  Writes one JSON object.

    BPMSM_LIB=bulletproofs_amd/libbpmsm_opcount.so python scripts/opcount.py > profiles/r03_opcount.json   (bench.py reads the newest)
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BPMSM_LIB", os.path.join(ROOT, "bulletproofs_amd", "libbpmsm_opcount.so"))

MAD_PER_MUL, MAD_PER_SQR = 102, 65


def read(L, name, reset=True):
    f = getattr(L, name)
    f.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    f.restype = ctypes.c_int
    m, q = ctypes.c_ulonglong(), ctypes.c_ulonglong()
    assert f(ctypes.byref(m), ctypes.byref(q), 1 if reset else 0) == 0
    return m.value, q.value


def heartbeat():
    """the instrumented build is slow (one atomic per field operation): keep the harness's silence watchdog fed"""
    import threading, time
    t0 = time.time()
    def beat():
        while True:
            time.sleep(20)
            print("[opcount] %.0f s" % (time.time() - t0), file=sys.stderr, flush=True)
    threading.Thread(target=beat, daemon=True).start()


def main():
    heartbeat()
    import torch
    import bulletproofs_amd as bpa
    import bench
    L = bpa.lib()
    dev = torch.device("cuda", 0)
    out = {"mad_per_fp_mul": MAD_PER_MUL, "mad_per_fp_sqr": MAD_PER_SQR}
    ctx = bpa.Context(0)
    golden = bench.load_batch()
    g_xy, h_xy, G_xy, H_xy = (bytes.fromhex(golden[x]) for x in ("g_vec", "h_vec", "G", "H"))
    P0 = int(os.environ.get("OPCOUNT_PROOFS", "65536"))  # the headline launch shape (count where the bench runs)
    skip_verify = bool(os.environ.get("OPCOUNT_SKIP_VERIFY"))  # debugging aid: straight to the raw MSM
    for wb, lanes in (() if skip_verify else ((0, 2), (0, 16), (16, 2))):
        P = P0 if (wb, lanes) == (0, 2) else min(P0, 8192)
        print("[opcount] table window %d, lanes %d, %d proofs" % (wb, lanes, P), file=sys.stderr, flush=True)
        gens = bpa.Gens(ctx, g_xy, h_xy, G_xy, H_xy, wb)
        print("[opcount]   generators built", file=sys.stderr, flush=True)
        proofs, coms, plen = bench.make_distinct_batch(ctx, gens, G_xy, H_xy, 0, 1, P, b"opcount")
        print("[opcount]   proofs made", file=sys.stderr, flush=True)
        d_p = torch.frombuffer(bytearray(proofs), dtype=torch.uint8).to(dev)
        d_c = torch.frombuffer(bytearray(coms), dtype=torch.uint8).to(dev)
        d_v = torch.full((P,), 77, dtype=torch.uint8, device=dev)
        ctx.set_verify_lanes(lanes)
        ctx.sync()
        read(L, "bp_opcount_verify")
        ctx.rangeproof_verify_batch_device(gens, 64, 1, d_p.data_ptr(), plen, d_c.data_ptr(), P, d_v.data_ptr())
        ctx.sync()
        assert not bool(d_v.any().item())
        mul, sqr = read(L, "bp_opcount_verify")
        # verify_kernels.hip holds k_bpv_decompress (one sqrt per proof point) and k_bpv_msm; the decompression's share is
        # counted separately below and subtracted
        key = "k_bpv_msm_table%s_lanes%d" % ("auto" if not wb else wb, lanes)
        out[key] = {"proofs": P, "fp_mul": mul, "fp_sqr": sqr, "table_bytes": gens.table_bytes, "includes": "k_bpv_decompress + k_bpv_msm"}
        gens.close()
    # decompression alone: a batch of malformed-at-the-first-point proofs still decompresses every point but skips the MSM
    if skip_verify:
        return msm_part(ctx, L, out, torch, bench, dev)
    print("[opcount] decompression alone", file=sys.stderr, flush=True)
    gens = bpa.Gens(ctx, g_xy, h_xy, G_xy, H_xy, 8)
    proofs, coms, plen = bench.make_distinct_batch(ctx, gens, G_xy, H_xy, 0, 1, 1024, b"opcount-d")
    pb = bytearray(proofs)
    for j in range(1024):
        pb[j * plen] = 5
    d_p = torch.frombuffer(pb, dtype=torch.uint8).to(dev)
    d_c = torch.frombuffer(bytearray(coms), dtype=torch.uint8).to(dev)
    d_v = torch.full((1024,), 77, dtype=torch.uint8, device=dev)
    ctx.sync()
    read(L, "bp_opcount_verify")
    ctx.rangeproof_verify_batch_device(gens, 64, 1, d_p.data_ptr(), plen, d_c.data_ptr(), 1024, d_v.data_ptr())
    ctx.sync()
    assert set(d_v.cpu().tolist()) == {255}
    mul, sqr = read(L, "bp_opcount_verify")
    out["k_bpv_decompress"] = {"proofs": 1024, "fp_mul": mul, "fp_sqr": sqr}
    gens.close()
    dmul, dsqr = mul / 1024, sqr / 1024
    for k, v in list(out.items()):
        if k.startswith("k_bpv_msm"):
            pm, ps = v["fp_mul"] / v["proofs"] - dmul, v["fp_sqr"] / v["proofs"] - dsqr
            v["msm_only_fp_mul_per_proof"] = pm
            v["msm_only_fp_sqr_per_proof"] = ps
            v["executed_mad32_per_proof"] = pm * MAD_PER_MUL + ps * MAD_PER_SQR
    ctx.set_verify_lanes(0)
    msm_part(ctx, L, out, torch, bench, dev)


def msm_part(ctx, L, out, torch, bench, dev):
    for log2n in (int(os.environ.get("OPCOUNT_MSM_LOG2", "20")),):
        N = 1 << log2n
        print("[opcount] raw MSM 2^%d: points" % log2n, file=sys.stderr, flush=True)
        pts = bytearray()
        for s0 in range(0, N, 1 << 16):
            pts += ctx.fixed_base_mul_batch(bench.scalars_in_range(b"bpmsm-c5-r-%d" % s0, 1 << 16))
        scal = bench.shake(b"bpmsm-c5-k", 32 * N)
        d_pts = torch.frombuffer(pts, dtype=torch.uint8).to(dev)
        d_sc = torch.frombuffer(bytearray(scal), dtype=torch.uint8).to(dev)
        torch.cuda.synchronize()
        print("[opcount] raw MSM 2^%d: run" % log2n, file=sys.stderr, flush=True)
        read(L, "bp_opcount_msm_sorted")
        read(L, "bp_opcount_msm_kernels")
        res = ctx.msm_device(d_pts.data_ptr(), d_sc.data_ptr(), N)
        m1, q1 = read(L, "bp_opcount_msm_sorted")
        m2, q2 = read(L, "bp_opcount_msm_kernels")
        out["msm_2p%d" % log2n] = {"points": N, "fp_mul": m1 + m2, "fp_sqr": q1 + q2, "result": res.hex(),
                                  "executed_mad32_per_point": ((m1 + m2) * MAD_PER_MUL + (q1 + q2) * MAD_PER_SQR) / N,
                                  "sorted_tu": {"fp_mul": m1, "fp_sqr": q1}, "msm_kernels_tu": {"fp_mul": m2, "fp_sqr": q2}}
    ctx.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
