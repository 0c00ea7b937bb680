#!/bin/bash
# (BP_TUNING build) lanes per window group at 2^20 / 2^22 and window width at 2^16..2^19
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
t() { env "$@" timeout -k 10 100 python3 $R/scripts/experiments/msm_once.py $K 10 2>/dev/null | grep "^msm" | tail -6 | awk '{s+=$3} END {printf "%.3f", s/NR}'; }
for l in default 512,512,256 512,384,384 640,384,256 512,256,256 768,512,256 384,384,256 512,384,512; do
  for K in 20 22; do
    if [ "$l" = default ]; then e=A=1; else e=BP_MSM_LANES=$((${l%%,*}*256)),$(echo $l | cut -d, -f2 | awk '{print $1*256}'),$(echo $l | cut -d, -f3 | awk '{print $1*256}'); fi
    echo "lanes $l 2^$K: $(t $e) ms"
  done
done
for c in 0 11 12 13 14 15; do
  for K in 16 17 18 19; do
    if [ $c = 0 ]; then e=A=1; else e=BP_MSM_C=$c; fi
    echo "c=$c 2^$K: $(t $e) ms"
  done
done
