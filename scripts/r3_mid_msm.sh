#!/bin/bash
# mid-size raw MSMs (2^16..2^19): window groups and quad / single-lane reductions (BP_TUNING build)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # label, env...
  lbl=$1; shift
  for k in 16 17 18 19 20; do
    t=$(env "$@" timeout -k 10 100 python3 $R/scripts/experiments/msm_once.py $k 8 2>/dev/null | grep "^msm" | tail -4 | awk '{s+=$3} END {printf "%.3f", s/NR}') || exit 1
    echo "$lbl 2^$k: $t ms"
  done
}
run "default        " A=1
run "RC_T=64,64     " BP_MSM_RC_T=64,64
run "groups 9       " BP_MSM_GROUPS=9
run "groups 8,1 quad" BP_MSM_GROUPS=8,1 BP_MSM_RC_T=64,64
run "groups 6,3 quad" BP_MSM_GROUPS=6,3 BP_MSM_RC_T=64,64
run "groups 4,4,1 q " BP_MSM_GROUPS=4,4,1 BP_MSM_RC_T=64,64
