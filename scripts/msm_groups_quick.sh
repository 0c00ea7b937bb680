#!/bin/bash
# unprofiled 2^20 MSM timings for several window-group plans (BP_MSM_GROUPS)
cd /tmp && export TMPDIR=/tmp
for g in $1; do
  export BP_MSM_GROUPS=$g
  echo "== groups $g"; timeout -k 10 100 python3 $GRAFT_REPO_ROOT/scripts/experiments/msm_once.py ${2:-20} 8 2>&1 | grep "^msm" | tail -4 || exit 1
done
