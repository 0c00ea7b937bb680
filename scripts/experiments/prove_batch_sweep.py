"""Batch-prover throughput for the golden C1 / C2 / C4 fixtures (BP_TUNING build: BP_PROVE_LANES / BP_PROVE_CHUNK sweeps)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bulletproofs_amd as bpa
def be(xs):
    return b"".join(int(x, 16).to_bytes(32, "big") for x in xs)
def load(name, idx=None):
    d = json.load(open(os.path.join(ROOT, "tests", "golden", name)))
    return d if idx is None else d[idx]
which = sys.argv[1] if len(sys.argv) > 1 else "c1"
case, kind, nb = {"c1": (load("bp_cases.json", 3), 0, 2048), "c2": (load("bp_c2_n64_m16.json"), 0, 256), "c4": (load("bpplus_c4_n64_m64.json"), 1, 64)}[which]
if len(sys.argv) > 2:
    nb = int(sys.argv[2])
ctx = bpa.Context(0)
g, h, G, H = bpa.generate_bases(kind, bpa.KZEN_SEED, case["n"], case["m"])
gens = bpa.Gens(ctx, g, h, G, H, 0)
sec, bl, tape = be(case["secrets"]), be(case["blindings"]), bytes.fromhex(case["tape"])
pb = ctx.wip_rangeproof_prove_batch if kind else ctx.rangeproof_prove_batch
pb(gens, case["n"], case["m"], sec * 16, bl * 16, tape * 16, len(tape))
for rep in range(3):
    t0 = time.perf_counter()
    got = pb(gens, case["n"], case["m"], sec * nb, bl * nb, tape * nb, len(tape))
    dt = time.perf_counter() - t0
    assert len(got) == nb and all(p.hex() == case["proof"] for p in got[:3] + got[-3:])
    print("%s nb=%d: %.1f proofs/s (%.2f ms)" % (which, nb, nb / dt, dt * 1e3), flush=True)
