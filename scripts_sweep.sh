#!/bin/bash
# usage: scripts_sweep.sh "<P list>" "<L list>" [extra bench args]  -- one summary line per run
PS=${1:-"4096 16384"}; LS=${2:-"16"}; shift 2
for P in $PS; do for L in $LS; do
timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --proofs-per-gpu $P --lanes $L "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('P=$P L=$L', round(d['value']), {k:round(v,3) for k,v in d['kernel_ms'].items()}, round(d['roofline']['frac'],4))" || exit 1
done; done
