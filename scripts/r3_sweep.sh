#!/bin/bash
# usage: r3_sweep.sh VAR "v1 v2 ..." [log2]   -> unprofiled 2^k MSM timings per value of the env knob
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in $2; do
  export $1=$v
  echo "== $1=$v 2^${3:-20}"; timeout -k 10 200 python3 $R/scripts/experiments/msm_once.py ${3:-20} 8 2>&1 | grep "^msm" | tail -3 || exit 1
done
