#!/bin/bash
# usage: scripts/msm_sizes.sh "<log2 n list>"  -> ms per MSM for the LDS slice path (1) and the sorted path (2)
for lg in $1; do for path in 1 2; do
  echo -n "n=2^$lg path=$path: "
  BP_MSM_PATH=$path python3 bench.py --workload msm --msm-log2 $lg --steps 10 --warmup 2 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '
  echo
done; done
