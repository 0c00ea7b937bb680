#!/bin/bash
# lanes per proof of the fixed-generator kernel (BP_FIX_LANES) by batch size: ms per call one at a time / in flight
cd /tmp && export TMPDIR=/tmp
for tf in 0 4 8 16 32 64; do
  echo "== BP_FIX_LANES=$tf"
  BP_FIX_LANES=$tf timeout -k 10 200 python3 $GRAFT_REPO_ROOT/scripts/experiments/mid_batch.py 2>&1 | grep "^n=\|differ" || exit 1
done
