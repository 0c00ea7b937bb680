#!/bin/bash
# MSM A/B helper (run through gpurun): MSM parity tests, kernel timeline of one 2^20 call and unprofiled call times
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_msm.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2
bash scripts/msm_timeline.sh "${1:-5,3,1}" 2>&1 | grep -v "^\["
