import hashlib, threading, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bulletproofs_amd as bpa
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
c1, c2 = bpa.Context(0), bpa.Context(0)
def mk(ctx, n):
    sc = b"".join(hashlib.sha256(b"r%d" % i).digest() for i in range(n))
    pts = b"".join(ctx.fixed_base_mul_batch(sc[i:i + 32 * 65536]) for i in range(0, len(sc), 32 * 65536))
    ks = b"".join(hashlib.sha256(b"k%d" % i).digest() for i in range(n))
    return torch.frombuffer(bytearray(pts), dtype=torch.uint8).to(dev), torch.frombuffer(bytearray(ks), dtype=torch.uint8).to(dev), n
big = mk(c1, 1 << 20); small = mk(c2, 128)
torch.cuda.synchronize()
def loop(ctx, d, reps, out):
    ctx.msm_device(d[0].data_ptr(), d[1].data_ptr(), d[2])
    t0 = time.perf_counter()
    for _ in range(reps): ctx.msm_device(d[0].data_ptr(), d[1].data_ptr(), d[2])
    out.append((time.perf_counter() - t0) / reps * 1e3)
a, b = [], []
loop(c1, big, 20, a); loop(c2, small, 60, b)
print("alone: big %.2f ms, small %.2f ms" % (a[0], b[0]))
a, b = [], []
t1 = threading.Thread(target=loop, args=(c1, big, 20, a)); t2 = threading.Thread(target=loop, args=(c2, small, 60, b))
t1.start(); t2.start(); t1.join(); t2.join()
print("together: big %.2f ms, small %.2f ms" % (a[0], b[0]))
