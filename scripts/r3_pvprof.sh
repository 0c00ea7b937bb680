#!/bin/bash
# kernel traces of the device batch prover; args: "<case> <nb>" pairs, e.g. "c1 2048" "c2 512" "c4 256"; LANES=1 so the steps do not interleave
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in "$@"; do
  set -- $w
  BP_PROVE_LANES=${LANES:-1} timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3_pv_prof_$1 -o pv -- python3 $R/scripts/experiments/prove_batch_sweep.py $1 $2 > $R/gpurun_out/r3_pv_prof_$1.log 2>&1 || exit 1
  grep "proofs/s" $R/gpurun_out/r3_pv_prof_$1.log
done
