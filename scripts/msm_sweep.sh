#!/bin/bash
# MSM tuning sweep on the GPU box (result hex must not change): edit the run lines
set -o pipefail
run() { echo "== $*"; env "$@" timeout -k 10 120 python bench.py --workload msm --steps 40 --warmup 3 --msm-log2 ${LOG2:-20} 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['value'], d['config']['result'][:16], d['two_in_flight']['value'])
    elif 'amdgpu.ids' not in l: print(l.rstrip()[:300])
" || exit 1; }
run BP_X=1
run BP_X=2
run BP_MSM_GROUPS=99
export LOG2=22
run BP_X=1
