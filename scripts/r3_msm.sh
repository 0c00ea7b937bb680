#!/bin/bash
# round 3: MSM tests + timeline + sizes.  usage: r3_msm.sh [tests|notests] "<log2 sizes>" "<groups for timeline>"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
if [ "$1" = "tests" ]; then
  cd $R && timeout -k 10 900 python3 -u -m pytest tests/test_gpu_msm.py -x -q --timeout 300 --timeout-method thread -m gpu > $O/r3_msm_tests.log 2>&1 || { tail -30 $O/r3_msm_tests.log; exit 1; }
  tail -3 $O/r3_msm_tests.log
  cd /tmp
fi
for k in $2; do
  echo "== 2^$k (unprofiled)"; timeout -k 10 200 python3 $R/scripts/experiments/msm_once.py $k 8 2>&1 | grep "^msm" | tail -4 || exit 1
done
for g in $3; do
  if [ "$g" != "default" ]; then export BP_MSM_GROUPS=$g; fi
  rm -rf /tmp/tl
  timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/tl -o tl --output-format csv -- python3 $R/scripts/experiments/msm_once.py ${TLK:-20} 4 > /tmp/o.log 2>&1 || { tail -20 /tmp/o.log; exit 1; }
  echo "== groups $g 2^${TLK:-20} (profiled)"; python3 $R/scripts/experiments/timeline.py /tmp/tl
done
