#!/bin/bash
# window-group plans at 2^20 / 2^22 with quad reductions everywhere (BP_TUNING build)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for g in default 6,2,1 4,4,1 7,1,1 6,3 7,2 8,1 4,3,2 3,3,2,1 5,2,2 4,2,2,1; do
  for k in 20 22; do
    if [ "$g" = default ]; then e=A=1; else e=BP_MSM_GROUPS=$g; fi
    t=$(env $e timeout -k 10 100 python3 $R/scripts/experiments/msm_once.py $k 10 2>/dev/null | grep "^msm" | tail -6 | awk '{s+=$3} END {printf "%.3f", s/NR}') || exit 1
    echo "groups $g 2^$k: $t ms"
  done
done
