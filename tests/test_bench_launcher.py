"""`bench.py --gpus N` starts N ranks itself (the flag used to be dead code).  BENCH_REHEARSAL=dry runs the
launcher, the rendezvous, the barriers, the max-over-ranks reduction and the JSON line with no library call, so it runs
on a box without a GPU; the measurement itself is covered on the GPU box."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("workload", ["verify", "msm"])
def test_gpus_flag_starts_that_many_ranks(workload):
    env = dict(os.environ, BENCH_REHEARSAL="dry")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", workload],
                         env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout       # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["dry_run"] is True


def test_single_gpu_run_does_not_spawn():
    env = dict(os.environ, BENCH_REHEARSAL="dry")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads(out.stdout.strip().splitlines()[-1])["n_gpus"] == 1
