"""Batches in flight: 65 536 distinct n=64 m=1 proofs, K asynchronous steps spread over 1-3 contexts (own streams and
workspaces, shared generator table) on the same GPU, with and without the library's own two-lane overlap
(bp_verify_set_overlap)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import bulletproofs_amd as bpa

P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
golden = bench.load_batch()
ctxs = [bpa.Context(0) for _ in range(3)]
g_xy, h_xy, G_xy, H_xy = (bytes.fromhex(golden[x]) for x in ("g_vec", "h_vec", "G", "H"))
gens = bpa.Gens(ctxs[0], g_xy, h_xy, G_xy, H_xy, 0)
proofs, coms, plen = bench.make_distinct_batch(ctxs[0], gens, G_xy, H_xy, 0, 1, P, b"two-ctx")
d_proofs = torch.frombuffer(bytearray(proofs), dtype=torch.uint8).to("cuda")
d_coms = torch.frombuffer(bytearray(coms), dtype=torch.uint8).to("cuda")
d_v = [torch.full((P,), 77, dtype=torch.uint8, device="cuda") for _ in range(6)]
torch.cuda.synchronize()

def run(nctx, n):
    for i in range(n):
        ctxs[i % nctx].rangeproof_verify_batch_device(gens, 64, 1, d_proofs.data_ptr(), plen, d_coms.data_ptr(), P, d_v[i % 6].data_ptr())
    for c in ctxs:
        c.sync()

for nctx, ov in ((1, 0), (1, 1), (2, 0), (2, 1), (3, 0), (3, 1), (1, 0), (1, 1)):
    for c in ctxs:
        c.set_verify_overlap(ov)
    run(nctx, 6)
    t0 = time.perf_counter()
    run(nctx, K)
    dt = time.perf_counter() - t0
    ok = all(bytes(v.cpu().numpy().tobytes()) == bytes(P) for v in d_v)
    print("%d context(s), overlap %d: %d steps of %d proofs: %.3f ms/step  %.3f M proofs/s  verdicts ok=%s" % (nctx, ov, K, P, dt / K * 1e3, P * K / dt / 1e6, ok), flush=True)
