#!/bin/bash
# quad lanes per output of the row/column sums (BP_MSM_RC_T="<last group>,<earlier groups>"), 2^19..2^22 (BP_TUNING build)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {
  lbl=$1; shift
  for k in 19 20 22; do
    t=$(env "$@" timeout -k 10 100 python3 $R/scripts/experiments/msm_once.py $k 8 2>/dev/null | grep "^msm" | tail -4 | awk '{s+=$3} END {printf "%.3f", s/NR}') || exit 1
    echo "$lbl 2^$k: $t ms"
  done
}
run "RC_T=64,64  " BP_MSM_RC_T=64,64
run "RC_T=64,32  " BP_MSM_RC_T=64,32
run "RC_T=64,128 " BP_MSM_RC_T=64,128
run "RC_T=128,128" BP_MSM_RC_T=128,128
run "RC_T=32,32  " BP_MSM_RC_T=32,32
run "RC_T=64,0   " BP_MSM_RC_T=64,0
