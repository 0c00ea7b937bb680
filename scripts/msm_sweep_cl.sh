#!/bin/bash
# usage: scripts/msm_sweep_cl.sh "<c list>" "<L list>" [log2n]   -> per-kernel average times of the raw MSM bench under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LG=${3:-20}
for c in $1; do for L in $2; do
  O=gpurun_out/msweep/c${c}_L${L}_n${LG}; mkdir -p $O
  BP_MSM_C=$c BP_MSM_L=$L timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $O -o kt -- python3 bench.py --workload msm --msm-log2 $LG --steps 3 --warmup 1 > $O/bench.json 2> $O/err.txt || exit 1
  python3 profiles/summarize.py stats $O/kt_results.db $O/stats.csv
  echo "== c=$c L=$L n=2^$LG: $(cut -c1-120 $O/bench.json | grep -o '"value": [0-9.]*')"
  grep -E "k_srt|horner|glv" $O/stats.csv | awk -F, '{printf "   %-28s %9.1f us\n",$1,$4}'
done; done
