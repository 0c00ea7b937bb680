#!/bin/bash
# batch-prover parity tests, then throughput by lane count for the aggregated shapes (BP_TUNING build)
set -o pipefail
timeout -k 10 400 python -u -m pytest tests/test_gpu_proto.py -x -q --timeout 180 --timeout-method thread -k "prove_batch" || exit 1
for lanes in 1 2 4; do
  echo "== lanes $lanes: c2 512, c4 256, c1 4096"
  BP_PROVE_LANES=$lanes timeout -k 10 120 python -u scripts/experiments/prove_batch_sweep.py c2 512 | tail -1 || exit 1
  BP_PROVE_LANES=$lanes timeout -k 10 120 python -u scripts/experiments/prove_batch_sweep.py c4 256 | tail -1 || exit 1
  BP_PROVE_LANES=$lanes timeout -k 10 120 python -u scripts/experiments/prove_batch_sweep.py c1 4096 | tail -1 || exit 1
done
