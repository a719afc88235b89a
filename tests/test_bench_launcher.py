"""CPU: `python bench.py --gpus N` from a plain shell must start its own N ranks (the driver may run it without
torch.distributed.run).  The rehearsal mode replaces the tokenizer launch by a host-side stand-in and runs the rest of
the multi-rank flow over gloo: rendezvous on 127.0.0.1, double-buffered async all-reduce of the [K] histogram, barriers,
MAX over ranks, ONE line from rank 0, exit status relayed by the parent."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _run(extra_args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"LIPVQ_BENCH_BACKEND": "gloo", "OMP_NUM_THREADS": "1"})
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + extra_args, env=env, cwd=str(ROOT),
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


@pytest.mark.parametrize("world", [2, 3])
def test_plain_invocation_spawns_ranks_and_relays_one_line(world):
    r = _run(["--gpus", str(world), "--steps", "4", "--warmup", "1", "--rehearse-launcher"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["rehearsal"] is True and "value" not in out and "metric" not in out
    assert out["n_gpus"] == world and out["world_size"] == world and out["backend"] == "gloo"
    assert out["steps"] == 4 and out["warmup"] == 1
    # every step's histogram was reduced over ALL ranks exactly once (rank r contributes r + 1 per step)
    assert out["usage_sum"] == out["expected_usage_sum"] == 5 * world * (world + 1) // 2
    # default = strong scaling (BASELINE config 4 as SURVEY 8d defines it): the ranks' shards add up to the workload's B sequences
    assert out["scaling"] == "strong" and out["sequences_per_step_all_ranks"] == 4096


def test_weak_scaling_keeps_the_per_gpu_batch():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--rehearse-launcher", "--scaling", "weak"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["scaling"] == "weak" and out["sequences_per_step_all_ranks"] == 2 * 4096


def test_child_failure_is_relayed_as_exit_status():
    # an external launcher that disagrees with --gpus is an error in the rank, and the rank's status is ours
    r = _run(["--gpus", "2", "--rehearse-launcher"], env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in r.stderr + r.stdout
