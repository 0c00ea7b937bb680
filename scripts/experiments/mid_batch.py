"""Mid-size batches: ms per call, one at a time and two in flight (run twice: with and without BP_VERIFY_NO_PAIR=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import bulletproofs_amd as bpa
golden = bench.load_batch()
ctx = bpa.Context(0)
g_xy, h_xy, G_xy, H_xy = (bytes.fromhex(golden[x]) for x in ("g_vec", "h_vec", "G", "H"))
gens = bpa.Gens(ctx, g_xy, h_xy, G_xy, H_xy, 0)
P = 32768
proofs, coms, plen = bench.make_distinct_batch(ctx, gens, G_xy, H_xy, 0, 1, P, b"mid")
pb = bytearray(proofs)
want = bench.tamper_batch(pb, plen, P, 4 * 33 + 64 + 31)
dp = torch.frombuffer(pb, dtype=torch.uint8).to("cuda"); dc = torch.frombuffer(bytearray(coms), dtype=torch.uint8).to("cuda")
dv = [torch.full((P,), 77, dtype=torch.uint8, device="cuda") for _ in range(2)]
for n in (2048, 4096, 8192, 12288, 16384, 24576, 32767):
    row = []
    for sync_each in (True, False):
        def run(reps):
            for i in range(reps):
                ctx.rangeproof_verify_batch_device(gens, 64, 1, dp.data_ptr(), plen, dc.data_ptr(), n, dv[i & 1].data_ptr())
                if sync_each:
                    ctx.sync()
            ctx.sync()
        run(4)
        t0 = time.perf_counter(); run(12); dt = (time.perf_counter() - t0) / 12
        for v in dv:
            assert bytes(v[:n].cpu().numpy().tobytes()) == bytes(want[:n]), "verdicts differ at n=%d" % n
        row.append(dt * 1e3)
    print("n=%5d  %.3f ms one at a time, %.3f ms in flight (%.2f M proofs/s)" % (n, row[0], row[1], n / row[1] / 1e3), flush=True)
