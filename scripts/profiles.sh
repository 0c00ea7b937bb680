#!/bin/bash
# Collects the per-round evidence under gpurun_out/prof (copied into profiles/ by hand afterwards).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof; rm -rf $O; mkdir -p $O
set -e
timeout -k 10 300 python3 bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err
# the same with successive calls one after the other: per-kernel durations with the GPU to themselves (roofline.avg_launch_ms)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt_serial -o kt -- python3 bench.py --no-cpu-baseline --no-overlap --no-rlc --no-bpplus --no-c2 --no-msm --no-tiled > $O/bench_serial_under_rocprof.json 2> $O/kt_serial.err
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/pmc_rd -o pmc -- python3 bench.py --no-cpu-baseline --no-rlc --no-bpplus --no-c2 --steps 3 --warmup 1 > $O/pmc_rd.json 2> $O/pmc_rd.err
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/pmc_wr -o pmc -- python3 bench.py --no-cpu-baseline --no-rlc --no-bpplus --no-c2 --steps 3 --warmup 1 > $O/pmc_wr.json 2> $O/pmc_wr.err
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $O/pmc_valu -o pmc -- python3 bench.py --no-cpu-baseline --no-rlc --no-bpplus --no-c2 --steps 3 --warmup 1 > $O/pmc_valu.json 2> $O/pmc_valu.err
timeout -k 10 300 python3 bench.py --workload msm > $O/msm_bench.json 2> $O/msm.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/msm_kt -o kt -- python3 bench.py --workload msm --steps 5 --warmup 2 > $O/msm_under_rocprof.json 2> $O/msm_kt.err
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/msm_pmc_rd -o pmc -- python3 bench.py --workload msm --steps 3 --warmup 1 > $O/msm_pmc_rd.json 2> $O/msm_pmc_rd.err
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/msm_pmc_wr -o pmc -- python3 bench.py --workload msm --steps 3 --warmup 1 > $O/msm_pmc_wr.json 2> $O/msm_pmc_wr.err
timeout -k 10 300 python3 bench.py --workload msm --msm-log2 22 --steps 5 > $O/msm_bench_2p22.json 2>> $O/msm.err
timeout -k 10 300 python3 bench.py --workload prove --steps 8 > $O/prove_latency.json 2> $O/prove.err
timeout -k 10 300 python3 scripts/experiments/bigtable_test.py 4096 12,14,16 > $O/bigtable.txt 2> $O/bigtable.err
timeout -k 10 120 ./bench_micro/ecops > $O/ecops.jsonl 2> $O/ecops.err
timeout -k 10 120 ./bench_micro/peak_mad32 > $O/peak_mad32.jsonl 2> $O/peak.err
# round 3: small-batch timeline and lane sweep, batch-prover rates and kernel breakdowns, raw-MSM sizes and timeline
bash scripts/small_batch_timeline.sh 512 > $O/small_batch_512.txt 2>&1
timeout -k 10 200 python3 -u scripts/experiments/slice8_test.py > $O/small_batch_lanes.txt 2> $O/slice8.err
for w in "c1 4096" "c1 16384" "c2 512" "c2 2048" "c4 256"; do timeout -k 10 200 python3 -u scripts/experiments/prove_batch_sweep.py $w >> $O/prove_batch.txt 2>> $O/prove_batch.err; done
LANES=1 bash scripts/r3_pvprof.sh "c1 2048" "c2 512" "c4 256" > $O/prove_batch_profiled.txt 2>&1
for w in c1 c2 c4; do echo "== $w (kernels from the last chunk's first launch on; two lanes interleave)" >> $O/prove_batch_kernels.txt; python3 scripts/pvprof_report.py gpurun_out/r3_pv_prof_$w/pv_results.db >> $O/prove_batch_kernels.txt; rm -rf gpurun_out/r3_pv_prof_$w; done
bash scripts/r3_msm.sh notests "16 18 20 22" "default" > $O/msm_sizes_timeline.txt 2>&1
cd $GRAFT_REPO_ROOT
for d in kt kt_serial pmc_rd pmc_wr pmc_valu msm_kt msm_pmc_rd msm_pmc_wr; do f=$(ls $O/$d/*_results.db); m=pmc; case $d in kt|kt_serial|msm_kt) m=stats;; esac; python3 profiles/summarize.py $m $f $O/$d.csv; rm -rf $O/$d; done  # raw rocprof databases stay on the box (gpurun merges back <= 64 MiB)
du -sh $O
echo PROFILES_OK
