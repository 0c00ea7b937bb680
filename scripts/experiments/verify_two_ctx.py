"""Batches in flight: 65 536 distinct n=64 m=1 proofs, K steps on one context with and without the overlapped scalar stage
(bp_verify_set_overlap), and K steps alternating between two contexts (own streams and workspaces, shared generator
table) on the same GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import bulletproofs_amd as bpa

P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
golden = bench.load_batch()
ctxs = [bpa.Context(0), bpa.Context(0)]
g_xy, h_xy, G_xy, H_xy = (bytes.fromhex(golden[x]) for x in ("g_vec", "h_vec", "G", "H"))
gens = bpa.Gens(ctxs[0], g_xy, h_xy, G_xy, H_xy, 0)
proofs, coms, plen = bench.make_distinct_batch(ctxs[0], gens, G_xy, H_xy, 0, 1, P, b"two-ctx")
d_proofs = torch.frombuffer(bytearray(proofs), dtype=torch.uint8).to("cuda")
d_coms = torch.frombuffer(bytearray(coms), dtype=torch.uint8).to("cuda")
d_v = [torch.full((P,), 77, dtype=torch.uint8, device="cuda") for _ in range(2)]
torch.cuda.synchronize()

def run(which, n):
    for i in range(n):
        j = which(i)
        ctxs[j].rangeproof_verify_batch_device(gens, 64, 1, d_proofs.data_ptr(), plen, d_coms.data_ptr(), P, d_v[j].data_ptr())
    for c in ctxs:
        c.sync()

for name, which, ov in (("one context, serial", lambda i: 0, 0), ("one context, overlap", lambda i: 0, 1), ("two contexts, serial", lambda i: i & 1, 0),
                        ("one context, serial", lambda i: 0, 0), ("one context, overlap", lambda i: 0, 1)):
    for c in ctxs:
        c.set_verify_overlap(ov)
    run(which, 3)
    t0 = time.perf_counter()
    run(which, K)
    dt = time.perf_counter() - t0
    ok = all(bytes(v.cpu().numpy().tobytes()) == bytes(P) for v in d_v[: (2 if name.startswith("two") else 1)])
    print("%-22s %d steps of %d proofs: %.3f ms/step  %.3f M proofs/s  verdicts ok=%s" % (name, K, P, dt / K * 1e3, P * K / dt / 1e6, ok), flush=True)
