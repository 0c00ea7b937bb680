#!/bin/bash
# RLC side measurement under the MSM's window-group plans, one and two contexts in flight
for g in "" 99; do for single in "" --rlc-single; do
  echo "== BP_MSM_GROUPS='$g' $single"
  BP_MSM_GROUPS=$g timeout -k 10 200 python bench.py --no-bpplus --no-cpu-baseline $single 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['rlc_batch']['value'], d['rlc_batch']['ms_per_step'], d['rlc_batch']['contexts_in_flight'])" || exit 1
done; done
