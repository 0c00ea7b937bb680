set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export BENCH_REHEARSAL=1
timeout -k 10 500 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-bpplus --no-c2 --no-rlc --no-tiled --no-sweep --no-msm --no-cpu-baseline > gpurun_out/rehearsal_verify.json 2> gpurun_out/rehearsal_verify.err
tail -c 1500 gpurun_out/rehearsal_verify.json
timeout -k 10 300 python3 bench.py --gpus 2 --workload msm --steps 5 --warmup 2 > gpurun_out/rehearsal_msm.json 2> gpurun_out/rehearsal_msm.err
tail -c 800 gpurun_out/rehearsal_msm.json
