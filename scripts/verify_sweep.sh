#!/bin/bash
# usage: scripts/verify_sweep.sh "<proofs-per-gpu list>" "<lanes list>" [extra bench args]  -> proofs/s and per-kernel ms
for P in $1; do for L in $2; do
  echo -n "P=$P lanes=$L: "
  python3 bench.py --no-cpu-baseline --proofs-per-gpu $P --lanes $L --steps 5 --warmup 2 $3 2>/dev/null | grep -o "\"value\": [0-9.]*\|kernel_ms[^}]*}" | tr '\n' ' '
  echo
done; done
