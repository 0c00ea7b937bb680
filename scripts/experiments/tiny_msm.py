"""Latency of tiny bp_msm calls (2, 8, 64 terms) through the C ABI."""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bulletproofs_amd as bpa
ctx = bpa.Context(0)
sc = lambda i: hashlib.sha256(b"tiny-%d" % i).digest()
pts = ctx.fixed_base_mul_batch(b"".join(sc(i) for i in range(200)))
for n in (1, 2, 8, 32, 64, 65, 128):
    p, s = pts[:64 * n], b"".join(sc(1000 + i) for i in range(n))
    ctx.msm(p, s, None)
    t0 = time.perf_counter()
    for _ in range(20):
        r = ctx.msm(p, s, None)
    print("n=%d: %.3f ms  %s" % (n, (time.perf_counter() - t0) / 20 * 1e3, r.hex()[:12]), flush=True)
