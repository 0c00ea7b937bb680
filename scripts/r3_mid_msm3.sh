#!/bin/bash
# one-group path (below 2^19 points): lanes per output of the row/column sums (BP_TUNING build)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for t in 64 32 16 0; do
  for k in 14 15 16 17 18; do
    v=$(BP_MSM_RC_T=$t,$t timeout -k 10 100 python3 $R/scripts/experiments/msm_once.py $k 10 2>/dev/null | grep "^msm" | tail -6 | awk '{s+=$3} END {printf "%.3f", s/NR}') || exit 1
    echo "RC_T=$t 2^$k: $v ms"
  done
done
