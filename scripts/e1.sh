set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for P in 512 4096; do
rm -rf /tmp/tl
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/tl -o tl --output-format csv -- python3 bench.py --no-cpu-baseline --no-rlc --no-bpplus --no-c2 --no-msm --no-sweep --no-tiled --proofs-per-gpu $P --steps 3 --warmup 2 > /tmp/o.log 2>&1 || { tail -20 /tmp/o.log; exit 1; }
echo "== P=$P"; grep "^{\"metric" /tmp/o.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['kernel_ms'])"
python3 scripts/experiments/timeline2.py /tmp/tl 14
done
