"""Small-batch verifier, four against eight lanes per proof-specific point (lanes = -4 / -8) by batch size: ms per call alone / in flight."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import bulletproofs_amd as bpa
golden = bench.load_batch()
ctx = bpa.Context(0)
g_xy, h_xy, G_xy, H_xy = (bytes.fromhex(golden[x]) for x in ("g_vec", "h_vec", "G", "H"))
gens = bpa.Gens(ctx, g_xy, h_xy, G_xy, H_xy, 0)
P = 1024
proofs, coms, plen = bench.make_distinct_batch(ctx, gens, G_xy, H_xy, 0, 1, P, b"thr8")
dp = torch.frombuffer(bytearray(proofs), dtype=torch.uint8).to("cuda"); dc = torch.frombuffer(bytearray(coms), dtype=torch.uint8).to("cuda")
dv = [torch.full((P,), 77, dtype=torch.uint8, device="cuda") for _ in range(2)]
for n in (64, 128, 256, 384, 512, 1024):
    row = []
    for lanes in (-4, -8):
        ctx.set_verify_lanes(lanes)
        for sync_each in (True, False):
            def run(reps):
                for i in range(reps):
                    ctx.rangeproof_verify_batch_device(gens, 64, 1, dp.data_ptr(), plen, dc.data_ptr(), n, dv[i & 1].data_ptr())
                    if sync_each:
                        ctx.sync()
                ctx.sync()
            run(4)
            t0 = time.perf_counter(); run(20); dt = (time.perf_counter() - t0) / 20
            assert bytes(dv[0][:n].cpu().numpy().tobytes()) == bytes(n)
            row.append(dt * 1e3)
    print("n=%5d  4 lanes/point: %.3f alone, %.3f in flight   8 lanes/point: %.3f / %.3f" % (n, row[0], row[1], row[2], row[3]), flush=True)
