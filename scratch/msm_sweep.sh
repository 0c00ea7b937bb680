#!/bin/bash
# MSM tuning sweep on the GPU box (result hex must not change)
set -o pipefail
run() { echo "== $*"; env "$@" timeout -k 10 120 python bench.py --workload msm --steps 30 --warmup 3 --msm-log2 ${LOG2:-20} 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['value'], d['config']['result'][:16], d['two_in_flight']['value'])
    elif 'amdgpu.ids' not in l: print(l.rstrip()[:300])
" || exit 1; }
export BP_MSM_PRESORT=0
run BP_MSM_GROUPS=6,3
run BP_MSM_GROUPS=7,2
run BP_MSM_GROUPS=6,2,1
run BP_MSM_GROUPS=5,3,1
run BP_MSM_GROUPS=4,3,2
run BP_MSM_GROUPS=6,3 BP_MSM_L=48
run BP_MSM_GROUPS=6,3 BP_MSM_L=40
run BP_MSM_GROUPS=6,3 BP_MSM_CH=32768
run BP_MSM_GROUPS=5,4 BP_MSM_L=48
export LOG2=22
run BP_MSM_GROUPS=9
run BP_MSM_GROUPS=5,4
run BP_MSM_GROUPS=6,3
run BP_MSM_GROUPS=4,3,2
export LOG2=18
run BP_MSM_GROUPS=99
run BP_MSM_GROUPS=5,4
run BP_MSM_GROUPS=6,3
