#!/bin/bash
# Kernel-level view of the default bench (per-proof verifier + RLC side measurement).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/rlcprof; rm -rf $O; mkdir -p $O
set -e
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 > $O/bench.json 2> $O/kt.err
python3 profiles/summarize.py stats $(ls $O/kt/*_results.db) $O/kt.csv
cat $O/kt.csv
