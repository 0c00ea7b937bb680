#!/bin/bash
# kernel timeline of the last batch-verification call of small_batch_once.py for two builds of the library
# usage: ab_timeline.sh KIND NPROOFS NKERNELS LIB_A LIB_B
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for L in $4 $5; do
  rm -rf /tmp/tl_$L
  export BPMSM_LIB=$R/bulletproofs_amd/$L OV=1
  timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/tl_$L -o tl --output-format csv -- python3 $R/scripts/experiments/small_batch_once.py $1 $2 > /tmp/o_$L.log 2>&1 || { tail -20 /tmp/o_$L.log; exit 1; }
  echo "== $L"; grep "^kind" /tmp/o_$L.log
  python3 $R/scripts/experiments/timeline2.py /tmp/tl_$L $3
done
