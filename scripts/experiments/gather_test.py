import torch, time
dev = torch.device("cuda", 0)
for gib in (4, 16, 57, 114):
    rows = gib * (1 << 30) // 64
    try:
        t = torch.empty((rows, 16), dtype=torch.int32, device=dev)
    except Exception as e:
        print(gib, "alloc failed", e); continue
    t.fill_(1)
    n = 1 << 26
    idx = torch.randint(0, rows, (n,), device=dev, dtype=torch.int64)
    out = torch.empty((n, 16), dtype=torch.int32, device=dev)
    torch.index_select(t, 0, idx, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        torch.index_select(t, 0, idx, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("table %d GiB: %.1f G gathers/s, %.2f TB/s read" % (gib, n / dt / 1e9, n * 64 / dt / 1e12), flush=True)
    del t, idx, out
    torch.cuda.empty_cache()
