"""One small batch-verification call alone, Bulletproofs (kind 0) or Bulletproofs+ (kind 1): wall time per call.
usage: small_batch_once.py KIND NPROOFS [M]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(__file__), "..", "..")))
import torch, bench
import bulletproofs_amd as bpa
kind, P = int(sys.argv[1]), int(sys.argv[2])
m = int(sys.argv[3]) if len(sys.argv) > 3 else 1
golden = bench.load_batch()
ctx = bpa.Context(0)
g_xy, h_xy, G_xy, H_xy = (bytes.fromhex(golden[x]) for x in ("g_vec", "h_vec", "G", "H"))
if m > 1:
    g_xy, h_xy = bpa.generate_bases(64 * m)[:2] if hasattr(bpa, "generate_bases") else (g_xy, h_xy)
gens = bpa.Gens(ctx, g_xy, h_xy, G_xy, H_xy, 0)
proofs, coms, plen = bench.make_distinct_batch(ctx, gens, G_xy, H_xy, kind, m, P, b"sb")
pb = bytearray(proofs)
for i in range(0, P, 7):  # every 7th proof tampered (a byte of the last scalar)
    pb[i * plen + plen - 1] ^= 1
dp = torch.frombuffer(pb, dtype=torch.uint8).to("cuda"); dc = torch.frombuffer(bytearray(coms), dtype=torch.uint8).to("cuda")
dv = torch.full((P,), 77, dtype=torch.uint8, device="cuda")
ctx.set_verify_overlap(int(os.environ.get("OV", "0")))
fn = ctx.wip_rangeproof_verify_batch_device if kind else ctx.rangeproof_verify_batch_device
ts = []
for i in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fn(gens, 64, m, dp.data_ptr(), plen, dc.data_ptr(), P, dv.data_ptr()); ctx.sync()
    ts.append((time.perf_counter() - t0) * 1e3)
v = dv.cpu().tolist()
ok = all((x != 0) == (i % 7 == 0) for i, x in enumerate(v))
print("kind %d P %d m %d: min %.3f ms median %.3f ms verdicts_ok %s" % (kind, P, m, min(ts[2:]), sorted(ts[2:])[5], ok), flush=True)
assert ok
