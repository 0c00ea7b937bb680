#!/bin/bash
# batch-prover parity tests, then throughput at several lane counts / batch sizes (BP_TUNING build)
set -o pipefail
timeout -k 10 400 python -u -m pytest tests/test_gpu_proto.py -x -q --timeout 180 --timeout-method thread -k "prove_batch" || exit 1
for lanes in 1 2 4; do
  echo "== lanes $lanes"
  BP_PROVE_LANES=$lanes timeout -k 10 120 python -u scripts/experiments/prove_batch_sweep.py c1 2048 || exit 1
done
echo "== c1 8192 (4 lanes)"; timeout -k 10 120 python -u scripts/experiments/prove_batch_sweep.py c1 8192 || exit 1
echo "== c1 16384 (4 lanes)"; timeout -k 10 120 python -u scripts/experiments/prove_batch_sweep.py c1 16384 || exit 1
echo "== c2 512"; timeout -k 10 120 python -u scripts/experiments/prove_batch_sweep.py c2 512 || exit 1
echo "== c2 2048"; timeout -k 10 120 python -u scripts/experiments/prove_batch_sweep.py c2 2048 || exit 1
echo "== trace c1 2048 lanes 1"; BP_PROVE_LANES=1 BP_PROVE_TRACE=1 timeout -k 10 120 python -u scripts/experiments/prove_batch_sweep.py c1 2048 2>&1 | tail -10
