#!/bin/bash
# Kernel-level view of the raw MSM as one window group (unshared kernel durations).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/msmprof; rm -rf $O; mkdir -p $O
set -e
export BP_MSM_GROUPS=${GROUPS_PLAN:-99}  # 99 = one window group: unshared kernel durations
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --workload msm --steps 5 --warmup 2 > $O/msm.json 2> $O/kt.err
python3 profiles/summarize.py stats $(ls $O/kt/*_results.db) $O/kt.csv
cat $O/kt.csv
cat $O/msm.json
