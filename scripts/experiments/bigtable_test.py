"""Does a fixed-base table of tens of GiB slow the gathers of k_bpv_msm down?  n=64 m=16 (np = 2050) proofs,
distinct challenges per proof (different tx bits), table windows 12 (5.9 GiB) / 14 / 16 (73 GiB)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bulletproofs_amd as bpa

case = json.load(open("tests/golden/bp_c2_n64_m16.json"))
n, m = 64, 16
g, h, G, H = bpa.generate_bases(0, bpa.KZEN_SEED, n, m)
ctx = bpa.Context(0)
proof = bytearray(bytes.fromhex(case["proof"]))
coms = b"".join(bytes.fromhex(c) for c in case["commitments"])
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
plen = len(proof)
pb = bytearray()
for j in range(P):
    p = bytearray(proof)
    for b in range(16):
        if (j >> b) & 1:
            p[4 * 33 + 64 + 31 - b // 8] ^= 1 << (b % 8)  # tx bits
    pb += p
dev = torch.device("cuda", 0)
d_p = torch.frombuffer(pb, dtype=torch.uint8).to(dev)
d_c = torch.frombuffer(bytearray(coms * P), dtype=torch.uint8).to(dev)
d_v = torch.zeros((P,), dtype=torch.uint8, device=dev)
for wb in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "12,14,16").split(",")]:
    t0 = time.perf_counter()
    gens = bpa.Gens(ctx, g, h, G, H, wb)
    tb = time.perf_counter() - t0
    ctx.set_verify_lanes(16)
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(4):
        ctx.rangeproof_verify_batch_device(gens, n, m, d_p.data_ptr(), plen, d_c.data_ptr(), P, d_v.data_ptr())
        ctx.sync()
    ms, cnt = ctx.profile_get(2)
    W = 256 // wb + 1
    adds = P * 2050 * W
    v = d_v.cpu().numpy()
    print("window %2d: table %.1f GiB built in %.2f s; k_bpv_msm %.2f ms per %d proofs; %.1f ps per table addition (incl. variable part); verdict[0]=%d" % (
        wb, gens.table_bytes / 2**30, tb, ms / cnt, P, ms / cnt * 1e9 / adds, v[0]), flush=True)
    ctx.profile_enable(False)
    gens.close()
