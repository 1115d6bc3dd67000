"""bench.py's launch contract: `--gpus N` without a launcher starts N ranks itself; a world size that differs from --gpus is refused (the
round-3 script parsed --gpus and never read it: a driver-run scaling point would have silently measured one GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def test_a_world_size_that_differs_from_gpus_is_refused():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "--gpus 2" in (p.stderr + p.stdout) and "1 rank" in (p.stderr + p.stdout)


@pytest.mark.gpu
def test_gpus_2_starts_two_ranks_by_itself():
    """both ranks on this GPU over gloo (LITEGS_BENCH_ONE_GPU: RCCL refuses two ranks on one device): the control flow of the N > 1 bench --
    self-launch, moment exchange, trained-state leg, bit-identical replicas -- on a small scene"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["LITEGS_BENCH_ONE_GPU"] = "1"
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "4", "--warmup", "2", "--config", "10k_400", "--frames", "2",
                        "--soak-steps", "8", "--no-cpu-baseline", "--no-operator-path", "--no-pmc", "--no-training-state"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 4
    assert out["dp_exchange"]["replicas_bit_identical"] is True
    assert out["dp_exchange"]["steady_state"]["after_steps"] == 8
