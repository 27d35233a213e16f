"""The RCCL path on ONE GPU: a 1-rank "nccl" process group in a fresh child process runs the two collectives of
the pupil-sharded step (moments all-reduce inside compute_rms2d, packed gradient all-reduce) on device tensors;
results must equal the group-less step bit for bit.  Also: bench.py --force-dist and the Adam loop with a group."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):      # the child builds its own 1-rank job
        env.pop(k, None)
    env["MASTER_ADDR"] = "127.0.0.1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cp = subprocess.run([sys.executable] + cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert cp.returncode == 0, f"{cmd} exited with {cp.returncode}\n{cp.stdout[-2000:]}\n{cp.stderr[-4000:]}"
    lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
    assert lines, cp.stdout[-2000:]
    return json.loads(lines[-1])


def test_one_rank_rccl_step_is_bitwise_the_plain_step():
    out = _run([os.path.join(ROOT, "tests", "rccl_child.py")])
    assert out["backend"] == "nccl" and out["world"] == 1 and out["n_ranks_seen"] == 1
    assert out["loss_bitwise_equal"], out
    assert out["grads_bitwise_equal"], out
    assert out["allgather_bitwise_equal"], out        # all-gather + fixed-order local sum: the same bits
    assert all(g > 0 for g in out["grad_norms"])


def test_bench_force_dist_runs_the_collectives():
    base = ["bench.py", "--steps", "5", "--warmup", "2", "--log2-pupil", "18", "--no-cpu-baseline", "--no-other-mode",
            "--no-also", "--no-graph-child", "--repeats", "1"]
    plain = _run(base)
    forced = _run(base + ["--force-dist"])
    assert forced["n_ranks_seen"] == 1 and forced["config"]["collectives"] == "rccl"
    assert plain["n_ranks_seen"] == 1 and plain["config"]["collectives"] == "none"
    assert forced["config"]["rms"] == plain["config"]["rms"]


def test_adam_loop_with_a_one_rank_rccl_group():
    a = _run(["examples/adam_loop.py", "--steps", "10", "--log2-pupil", "12"])
    b = _run(["examples/adam_loop.py", "--steps", "10", "--log2-pupil", "12", "--force-dist"])
    assert a["loss_final"] == b["loss_final"] and a["loss_initial"] == b["loss_initial"]
    assert b["loss_final"] != b["loss_initial"]
