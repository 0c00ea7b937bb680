set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_verify.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o p --output-format csv -- python3 bench.py --no-rlc --no-bpplus --no-c2 --no-msm --no-cpu-baseline --no-tiled --steps 10 > /tmp/o.log 2>&1 || { tail -20 /tmp/o.log; exit 1; }
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); grep -E "k_bpv_fix<|k_bpv_msm" $f | sed 's/(.*)"/"/' | cut -c1-120
grep "^{\"metric" /tmp/o.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['kernel_ms'], [(r['proofs_per_launch_per_gpu'], round(r['ms_per_launch'],3)) for r in j['batch_sweep']])"
