"""Mid-size batches: lanes per proof of the ladder kernel (bp_verify_set_lanes) by batch size, ms per call one at a time /
two in flight; '*' marks the automatic choice."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import bulletproofs_amd as bpa
golden = bench.load_batch()
ctx = bpa.Context(0)
g_xy, h_xy, G_xy, H_xy = (bytes.fromhex(golden[x]) for x in ("g_vec", "h_vec", "G", "H"))
gens = bpa.Gens(ctx, g_xy, h_xy, G_xy, H_xy, 0)
P = 49152
proofs, coms, plen = bench.make_distinct_batch(ctx, gens, G_xy, H_xy, 0, 1, P, b"midl")
dp = torch.frombuffer(bytearray(proofs), dtype=torch.uint8).to("cuda"); dc = torch.frombuffer(bytearray(coms), dtype=torch.uint8).to("cuda")
dv = [torch.full((P,), 77, dtype=torch.uint8, device="cuda") for _ in range(2)]
def auto(n):
    return 2 if n >= 65536 else 4 if n >= 32768 else 8 if n >= 12288 else 16
for n in (2048, 4096, 8192, 12288, 16384, 24576, 32768, 49152):
    out = []
    for lanes in (2, 4, 8, 16, 32):
        ctx.set_verify_lanes(lanes)
        row = []
        for sync_each in (True, False):
            def run(reps):
                for i in range(reps):
                    ctx.rangeproof_verify_batch_device(gens, 64, 1, dp.data_ptr(), plen, dc.data_ptr(), n, dv[i & 1].data_ptr())
                    if sync_each:
                        ctx.sync()
                ctx.sync()
            run(3)
            t0 = time.perf_counter(); run(10); row.append((time.perf_counter() - t0) / 10 * 1e3)
        assert bytes(dv[0][:n].cpu().numpy().tobytes()) == bytes(n)
        out.append("%s%d: %.2f/%.2f" % ("*" if lanes == auto(n) else "", lanes, row[0], row[1]))
    print("n=%5d  " % n + "   ".join(out), flush=True)
