"""Time and memory of the generator-table build (bp_gens_create, automatic window) for n=64 m=M."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(__file__), "..", "..")))
import torch, bench
import bulletproofs_amd as bpa
golden = bench.load_batch()
ctx = bpa.Context(0)
g_xy, h_xy, G_xy, H_xy = (bytes.fromhex(golden[x]) for x in ("g_vec", "h_vec", "G", "H"))
free0, total = torch.cuda.mem_get_info()
for i in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gens = bpa.Gens(ctx, g_xy, h_xy, G_xy, H_xy, 0)
    ctx.sync(); dt = time.perf_counter() - t0
    free1, _ = torch.cuda.mem_get_info()
    print("build %d: %.3f s, table %.1f GiB, device memory in use after the build %.1f GiB (before: %.1f)" % (
        i, dt, gens.table_bytes / 2**30, (total - free1) / 2**30, (total - free0) / 2**30), flush=True)
    del gens
