"""A few 2^20 MSMs through the C ABI (timeline runs under rocprofv3 --kernel-trace)."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bulletproofs_amd as bpa
log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N = 1 << log2
ctx = bpa.Context(0)
if os.environ.get("PIPELINE"):  # bp_msm_set_pipeline: 0 = groups from 2^19 points (default), 1 = one group, 2 = groups from 2^16 points
    ctx.set_msm_pipeline(int(os.environ["PIPELINE"]))
def stream(tag, count, start):
    return b"".join(hashlib.sha256(b"bpmsm-c5-%s-%d" % (tag, i)).digest() for i in range(start, start + count))
pts = bytearray()
for s0 in range(0, N, 1 << 16):
    pts += ctx.fixed_base_mul_batch(stream(b"r", min(1 << 16, N - s0), s0))
d_pts = torch.frombuffer(pts, dtype=torch.uint8).to("cuda")
d_sc = torch.frombuffer(bytearray(stream(b"k", N, 0)), dtype=torch.uint8).to("cuda")
torch.cuda.synchronize()
for i in range(reps):
    t0 = time.perf_counter()
    r = ctx.msm_device(d_pts.data_ptr(), d_sc.data_ptr(), N)
    print("msm %d: %.3f ms %s" % (i, (time.perf_counter() - t0) * 1e3, r.hex()[:16]), flush=True)
ctx.close()
