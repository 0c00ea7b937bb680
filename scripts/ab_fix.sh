#!/bin/bash
# A/B helper for the verifier kernels (run on the GPU box through gpurun): the full GPU parity suite, then the headline bench
# under rocprofv3 --kernel-trace --stats with the per-kernel averages of the MSM pair, the decompression and the sorted-MSM
# kernels, and the headline / sweep numbers of the JSON line.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o p --output-format csv -- python3 bench.py --no-rlc --no-bpplus --no-c2 --no-cpu-baseline --no-tiled --steps 10 > /tmp/o.log 2>&1 || { tail -20 /tmp/o.log; exit 1; }
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); grep -E "k_bpv_fix<4|k_bpv_msm<2|k_srt_acc|k_bpv_decompress|k_srt_rowcol2" $f | sed 's/(.*)"/"/' | cut -c1-120
grep "^{\"metric" /tmp/o.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['kernel_ms'], j['msm']['value'], [(r['proofs_per_launch_per_gpu'], round(r['ms_per_launch'],3)) for r in j['batch_sweep']])"
