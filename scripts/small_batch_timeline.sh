#!/bin/bash
# kernel timeline of ONE 512-proof (or $1-proof) batch verification, one call at a time
cd /tmp && export TMPDIR=/tmp
cat > /tmp/sb.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, bench
import bulletproofs_amd as bpa
P = int(sys.argv[1])
golden = bench.load_batch()
ctx = bpa.Context(0)
g_xy, h_xy, G_xy, H_xy = (bytes.fromhex(golden[x]) for x in ("g_vec", "h_vec", "G", "H"))
gens = bpa.Gens(ctx, g_xy, h_xy, G_xy, H_xy, 0)
proofs, coms, plen = bench.make_distinct_batch(ctx, gens, G_xy, H_xy, 0, 1, P, b"sb")
dp = torch.frombuffer(bytearray(proofs), dtype=torch.uint8).to("cuda"); dc = torch.frombuffer(bytearray(coms), dtype=torch.uint8).to("cuda")
dv = torch.full((P,), 77, dtype=torch.uint8, device="cuda")
ctx.set_verify_overlap(int(os.environ.get("OV", "0")))
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.rangeproof_verify_batch_device(gens, 64, 1, dp.data_ptr(), plen, dc.data_ptr(), P, dv.data_ptr()); ctx.sync()
    print("call %d: %.3f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)
PY
rm -rf /tmp/tl
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/tl -o tl --output-format csv -- python3 /tmp/sb.py ${1:-512} > /tmp/o.log 2>&1 || { tail -20 /tmp/o.log; exit 1; }
grep "^call" /tmp/o.log
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/tl/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last call = the last 10 kernels starting from a k_bpv_decompress
idx = [i for i, r in enumerate(rows) if 'k_bpv_decompress' in r['Kernel_Name']][-2 if len(__import__('sys').argv) < 2 else -1]
t0 = int(rows[idx]['Start_Timestamp'])
for r in rows[idx:]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print("%9.1f %9.1f %8.1f  q%-2s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:60]))
PY
