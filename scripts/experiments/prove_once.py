"""A few C1 proves through the C ABI (timeline runs under rocprofv3 --kernel-trace)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bulletproofs_amd as bpa
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# argv[2]: c1 (default) / c2 / c4 = the fixtures of BASELINE.json's configs[0], [1], [3]
CFG = sys.argv[2] if len(sys.argv) > 2 else "c1"
_f = {"c1": "bp_cases.json", "c2": "bp_c2_n64_m16.json", "c4": "bpplus_c4_n64_m64.json"}[CFG]
case = json.load(open(os.path.join(ROOT, "tests", "golden", _f)))
if CFG == "c1":
    case = case[3]
KIND = 1 if CFG == "c4" else 0
be = lambda xs: b"".join(int(x, 16).to_bytes(32, "big") for x in xs)
ctx = bpa.Context(0)
n, m = case["n"], case["m"]
g, h, G, H = bpa.generate_bases(KIND, bpa.KZEN_SEED, n, m)
gens = bpa.Gens(ctx, g, h, G, H, 0)
sec, bl, tape = be(case["secrets"]), be(case["blindings"]), bytes.fromhex(case["tape"])
import sys as _s
MODE = _s.argv[1] if len(_s.argv) > 1 else "prove"
coms = b"".join(bytes.fromhex(c) for c in case["commitments"])
for i in range(6):
    t0 = time.perf_counter()
    _prove = ctx.wip_rangeproof_prove if KIND else ctx.rangeproof_prove
    _verify = ctx.wip_rangeproof_verify if KIND else ctx.rangeproof_verify
    p = _prove(gens, n, sec, bl, tape) if MODE == "prove" else (_verify(gens, n, m, bytes.fromhex(case["proof"]), coms, 2) and b"") or bytes.fromhex(case["proof"])
    print("prove %d: %.3f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)
assert p.hex() == case["proof"]
