"""Soak of the overlapped device verifier: many asynchronous calls of random sizes / tamper patterns (BP and BP+ mixed with
host-pointer calls and MSMs on the same context), verdicts against the same calls with the overlap off."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import bulletproofs_amd as bpa

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
golden = bench.load_batch()
ctx = bpa.Context(0)
g_xy, h_xy, G_xy, H_xy = (bytes.fromhex(golden[x]) for x in ("g_vec", "h_vec", "G", "H"))
gens = bpa.Gens(ctx, g_xy, h_xy, G_xy, H_xy, 12)
P = 4096
proofs, coms, plen = bench.make_distinct_batch(ctx, gens, G_xy, H_xy, 0, 1, P, b"soak")
rng = random.Random(7)
calls = []
for r in range(rounds):
    n = rng.choice([1, 3, 17, 64, 200, 513, 1024, 2500, 4096])
    off = rng.randrange(0, P - n + 1)
    pb = bytearray(proofs[off * plen:(off + n) * plen])
    cb = bytearray(coms[off * 64:(off + n) * 64])
    want = [0] * n
    for j in range(n):
        t = rng.random()
        if t < 0.1:
            pb[j * plen + 4 * 33 + 64 + 31] ^= 1
            want[j] = 3
        elif t < 0.13:
            pb[j * plen + 33] = 9
            want[j] = 255
    calls.append((n, torch.frombuffer(pb, dtype=torch.uint8).to("cuda"), torch.frombuffer(cb, dtype=torch.uint8).to("cuda"), want, bytes(pb), bytes(cb)))
torch.cuda.synchronize()
pts = ctx.fixed_base_mul_batch(b"".join((i + 1).to_bytes(32, "big") for i in range(300)))
scal = b"".join((7 * i + 3).to_bytes(32, "big") for i in range(300))
msm_want = ctx.msm(pts, scal)
bad = 0
for ov in (1, 0, 1):
    ctx.set_verify_overlap(ov)
    outs = [torch.full((n,), 77, dtype=torch.uint8, device="cuda") for n, *_ in calls]
    torch.cuda.synchronize()
    for r, (n, dp, dc, want, pb, cb) in enumerate(calls):
        ctx.rangeproof_verify_batch_device(gens, 64, 1, dp.data_ptr(), plen, dc.data_ptr(), n, outs[r].data_ptr())
        if r % 23 == 5:
            assert ctx.rangeproof_verify_batch(gens, 64, 1, pb, cb) == want, "host-pointer call differs at %d" % r
        if r % 31 == 7:
            assert ctx.msm(pts, scal) == msm_want, "msm differs at %d" % r
        if r % 41 == 0:
            ctx.sync()
    ctx.sync()
    for r, (n, dp, dc, want, pb, cb) in enumerate(calls):
        got = outs[r].cpu().tolist()
        if got != want:
            bad += 1
            print("overlap %d call %d (n=%d): %d verdicts differ" % (ov, r, n, sum(a != b for a, b in zip(got, want))), flush=True)
    print("overlap %d: %d calls checked" % (ov, len(calls)), flush=True)
print("SOAK_OK" if not bad else "SOAK_FAILED %d" % bad)
