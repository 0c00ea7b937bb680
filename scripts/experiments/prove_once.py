"""A few C1 proves through the C ABI (timeline runs under rocprofv3 --kernel-trace)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bulletproofs_amd as bpa
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
case = json.load(open(os.path.join(ROOT, "tests", "golden", "bp_cases.json")))[3]
be = lambda xs: b"".join(int(x, 16).to_bytes(32, "big") for x in xs)
ctx = bpa.Context(0)
n, m = case["n"], case["m"]
g, h, G, H = bpa.generate_bases(0, bpa.KZEN_SEED, n, m)
gens = bpa.Gens(ctx, g, h, G, H, 0)
sec, bl, tape = be(case["secrets"]), be(case["blindings"]), bytes.fromhex(case["tape"])
import sys as _s
MODE = _s.argv[1] if len(_s.argv) > 1 else "prove"
coms = b"".join(bytes.fromhex(c) for c in case["commitments"])
for i in range(6):
    t0 = time.perf_counter()
    p = ctx.rangeproof_prove(gens, n, sec, bl, tape) if MODE == "prove" else (ctx.rangeproof_verify(gens, n, m, bytes.fromhex(case["proof"]), coms, 2) and b"") or bytes.fromhex(case["proof"])
    print("prove %d: %.3f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)
assert p.hex() == case["proof"]
