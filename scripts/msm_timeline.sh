#!/bin/bash
# usage: tl_run.sh "<groups> <groups> ..."  -> timelines + unprofiled timings
cd /tmp && export TMPDIR=/tmp
for g in $1; do
  export BP_MSM_GROUPS=$g
  rm -rf /tmp/tl
  timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/tl -o tl --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/experiments/msm_once.py 20 4 > /tmp/o.log 2>&1 || { tail -20 /tmp/o.log; exit 1; }
  echo "== groups $g (profiled)"; python3 $GRAFT_REPO_ROOT/scripts/experiments/timeline.py /tmp/tl
  echo "== groups $g (unprofiled)"; timeout -k 10 100 python3 $GRAFT_REPO_ROOT/scripts/experiments/msm_once.py 20 8 2>&1 | grep "^msm" | tail -3 || exit 1
done
