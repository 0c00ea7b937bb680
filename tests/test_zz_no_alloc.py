"""The no-allocation entry point (bp_rangeproof_verify_batch_device_noalloc): same verdict bytes as
bp_rangeproof_verify_batch_device, through the C ABI, on two calls in a row (the second smaller, so the fixed scratch
array still holds the first call's contents), then the size limit.  Collected last on purpose (file name)."""
import json
import os

import pytest

import bulletproofs_amd as bpa

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _batch(count, tamper_every, mal_at=()):
    with open(os.path.join(ROOT, "tests", "golden", "bp_c3_batch_n64_m1.json")) as f:
        b = json.load(f)
    plen = 292 + 66 * 6
    proofs, coms = bytes.fromhex(b["proofs"]), bytes.fromhex(b["commitments"])
    pb, cb = bytearray(), bytearray()
    for j in range(count):
        i = (j * 7 + 3) % b["count"]
        p = bytearray(proofs[i * plen:(i + 1) * plen])
        if j % tamper_every == 1:
            p[4 * 33 + 64 + 31] ^= 1      # tx
        if j in mal_at:
            p[0] = 5                      # not a SEC1 prefix: malformed
        pb += p
        cb += coms[i * 64:(i + 1) * 64]
    return b, plen, bytes(pb), bytes(cb)


def test_noalloc_matches_the_allocating_entry_point():
    import torch
    dev = torch.device("cuda", 0)
    with bpa.Context(0) as ctx:
        b, plen, pb, cb = _batch(700, 5, mal_at=(13, 400))
        gens = bpa.Gens(ctx, bytes.fromhex(b["g_vec"]), bytes.fromhex(b["h_vec"]), bytes.fromhex(b["G"]), bytes.fromhex(b["H"]), 12)
        try:
            for count in (700, 93):       # second call smaller: stale contents of the first stay in the array
                d_p = torch.frombuffer(bytearray(pb[:count * plen]), dtype=torch.uint8).to(dev)
                d_c = torch.frombuffer(bytearray(cb[:count * 64]), dtype=torch.uint8).to(dev)
                want = torch.full((count,), 77, dtype=torch.uint8, device=dev)
                got = torch.full((count,), 78, dtype=torch.uint8, device=dev)
                torch.cuda.synchronize()
                ctx.rangeproof_verify_batch_device(gens, 64, 1, d_p.data_ptr(), plen, d_c.data_ptr(), count, want.data_ptr())
                ctx.sync()
                ctx.rangeproof_verify_batch_device_noalloc(gens, 64, 1, d_p.data_ptr(), plen, d_c.data_ptr(), count, got.data_ptr())
                ctx.sync()
                w, g = want.cpu().tolist(), got.cpu().tolist()
                assert g == w
                assert set(w) == {0, 3, 255} if count == 700 else set(w) >= {0, 3}
                assert w[13] == 255 and w[1] == 3 and w[0] == 0
            # beyond the fixed scratch: a clear error, nothing launched
            big = bpa.noalloc_scratch_bytes() // 20000
            d_p = torch.zeros((big * plen,), dtype=torch.uint8, device=dev)
            d_c = torch.zeros((big * 64,), dtype=torch.uint8, device=dev)
            d_v = torch.full((big,), 9, dtype=torch.uint8, device=dev)
            with pytest.raises(bpa.BpError) as e:
                ctx.rangeproof_verify_batch_device_noalloc(gens, 64, 1, d_p.data_ptr(), plen, d_c.data_ptr(), big, d_v.data_ptr())
            assert e.value.code == bpa.BP_EINVAL and "split the batch" in str(e.value)
            assert set(d_v.cpu().tolist()) == {9}
        finally:
            gens.close()


def test_noalloc_calls_of_two_contexts_serialise():
    """Two contexts on one GPU share the fixed scratch array: their un-synchronised no-allocation calls are ordered on the
    device by the per-GPU event (each call's stream waits for the previous call's kernels), so both give the allocating
    entry point's verdicts instead of racing."""
    import torch
    dev = torch.device("cuda", 0)
    with bpa.Context(0) as ctx_a, bpa.Context(0) as ctx_b:
        b, plen, pb, cb = _batch(600, 4, mal_at=(7,))
        gens = bpa.Gens(ctx_a, bytes.fromhex(b["g_vec"]), bytes.fromhex(b["h_vec"]), bytes.fromhex(b["G"]), bytes.fromhex(b["H"]), 12)
        try:
            halves = [(0, 350), (350, 600)]
            d_p = [torch.frombuffer(bytearray(pb[lo * plen:hi * plen]), dtype=torch.uint8).to(dev) for lo, hi in halves]
            d_c = [torch.frombuffer(bytearray(cb[lo * 64:hi * 64]), dtype=torch.uint8).to(dev) for lo, hi in halves]
            want = [torch.full((hi - lo,), 77, dtype=torch.uint8, device=dev) for lo, hi in halves]
            got = [torch.full((hi - lo,), 78, dtype=torch.uint8, device=dev) for lo, hi in halves]
            torch.cuda.synchronize()
            for i, (lo, hi) in enumerate(halves):
                ctx_a.rangeproof_verify_batch_device(gens, 64, 1, d_p[i].data_ptr(), plen, d_c[i].data_ptr(), hi - lo, want[i].data_ptr())
                ctx_a.sync()
            for rep in range(3):  # back to back from two contexts, no synchronisation in between
                for i, (c, (lo, hi)) in enumerate(zip((ctx_a, ctx_b), halves)):
                    c.rangeproof_verify_batch_device_noalloc(gens, 64, 1, d_p[i].data_ptr(), plen, d_c[i].data_ptr(), hi - lo, got[i].data_ptr())
            ctx_a.sync()
            ctx_b.sync()
            for i in range(2):
                assert got[i].cpu().tolist() == want[i].cpu().tolist()
        finally:
            gens.close()
