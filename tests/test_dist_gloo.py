"""
The N > 1 path on CPU: 2 processes, gloo backend.  What is under test is the HOST logic of the
pupil-sharded data-parallel scheme (torchoptics_amd/dist.py + the `group=` branch of
compute_rms2d): shard ranges, per-rank pupil slices generated from indices, the differentiable
moment all-reduce (#1) and the packed gradient all-reduce (#2).  The kernels themselves cannot run
here, so the CPU oracle is patched in for `trace_skew` / the moment reduction -- as the checker's
stand-in only; on the GPU box the same code path runs the HIP kernels (bench.py --gpus N).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_R, N_THETA = 24, 17          # 408 pupil points: not divisible by 2 -> uneven shards


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_moments(x, y, ok):
    yd, okd = y[0].double(), ok[0].double()
    z = torch.zeros_like(yd.sum(dim=(1, 2)))
    return torch.stack((yd.sum(dim=(1, 2)), (okd * yd).sum(dim=(1, 2)), (okd * yd * yd).sum(dim=(1, 2)),
                        okd.sum(dim=(1, 2)), z, z, z, z), dim=1)


def _trace_loss(leaves_t, xy, group, n_per_field, patch=setattr):
    """Assemble (package host logic) -> trace (oracle) -> compute_rms2d (package, sharded)."""
    import yaml_free_lenses as L
    import torchoptics_amd as ta
    from torchoptics_amd import lens_modeling as lm, ops, ray_tracing as rt
    from oracle import trace_oracle as orc
    patch(rt, "trace_skew", lambda *a, mode=None, **k: orc.trace_skew(*a, **k))
    patch(ops.SpotMomentsFunction, "apply", staticmethod(_oracle_moments))
    d = L.PRESCRIPTIONS["cooke"]
    st = lm.Structure(stop_idx=np.array(d["stop_idx"]), sequence=np.array(d["sequence"]), default_device="cpu")
    lens = lm.Lens(st, *leaves_t)
    specs = lm.Specs(st, torch.tensor([16.0]), torch.tensor([np.deg2rad(35.0)], dtype=torch.float32))   # failure-heavy
    tr = ta.RayTracer(mode="circular", n_rays=(N_R, N_THETA), rel_fields=(0., 0.707, 1.), wavelengths=("C", "d", "F"),
                      default_device="cpu")
    x, y, cx, cy, ok, back = tr.trace_rays(specs, lens, xy=xy)
    return rt.compute_rms2d(x, y, ok, group=group, n_per_field=n_per_field), ok


def _leaves():
    import yaml_free_lenses as L
    d = L.PRESCRIPTIONS["cooke"]
    return [torch.tensor(d[k], dtype=torch.float64, requires_grad=True) for k in ("c", "t", "nd", "v")]


def _worker(rank, world, port, out_dir, collective="allreduce"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_default_dtype(torch.float64)
    torch.set_num_threads(1)
    from torchoptics_amd import dist as tl_dist, ray_tracing as rt
    tl_dist.set_collective(collective)
    total = N_R * N_THETA
    a, b = tl_dist.shard_range(total, rank, world)
    xy = rt.circle_index_range(N_R, N_THETA, a, b, "cpu")
    leaves = _leaves()
    loss, ok = _trace_loss(leaves, xy, dist.group.WORLD, total * 3)
    loss.backward()
    tl_dist.all_reduce_grads(leaves)
    torch.save(dict(loss=loss.detach(), grads=[p.grad for p in leaves], shard=(a, b), n_ok=int(ok.sum())),
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_shard_range_covers_everything_once():
    from torchoptics_amd.dist import shard_range
    for n in (0, 1, 7, 408, 1 << 24):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
@pytest.mark.parametrize("collective", ["allreduce", "allgather"])
def test_two_rank_sharded_loss_and_grads_equal_unsharded(tmp_path, monkeypatch, collective):
    """Both shapes of the two exchanges: one all-reduce each (default), or all-gather + local sum in rank order."""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), collective), nprocs=2, join=True)
    res = [torch.load(tmp_path / f"rank{r}.pt", weights_only=True) for r in range(2)]
    assert res[0]["shard"] == (0, 204) and res[1]["shard"] == (204, 408)

    # unsharded reference in this process: whole pupil, no group
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_default_dtype(torch.float64)
    try:
        from torchoptics_amd import ray_tracing as rt
        leaves = _leaves()
        xy = rt.circle_index_range(N_R, N_THETA, 0, N_R * N_THETA, "cpu")
        loss, ok = _trace_loss(leaves, xy, None, None, patch=monkeypatch.setattr)
        loss.backward()
    finally:
        torch.set_default_dtype(torch.float32)
    assert 0.3 < ok.double().mean() < 0.95                    # the case really has failed rays
    assert res[0]["n_ok"] + res[1]["n_ok"] == int(ok.sum())
    for r in res:
        assert abs(r["loss"].item() - loss.item()) < 1e-12 * abs(loss.item())
        for got, p in zip(r["grads"], leaves):
            assert torch.allclose(got, p.grad, rtol=1e-9, atol=1e-12)
    # every rank ends with bitwise identical gradients (packed fp64 all-reduce)
    for g0, g1 in zip(res[0]["grads"], res[1]["grads"]):
        assert torch.equal(g0, g1)


def test_spawn_local_ranks_starts_a_two_rank_job(tmp_path):
    """dist.spawn_local_ranks (what `bench.py --gpus N` and `adam_loop.py --gpus N` use when no launcher is around): a child
    torch.distributed.run with two ranks on 127.0.0.1, each joining through dist.init_group and summing its rank."""
    import subprocess
    script = tmp_path / "two_ranks.py"
    script.write_text(
        "import json, os, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import torch\n"
        "from torchoptics_amd import dist as tl_dist\n"
        "group = tl_dist.init_group('cpu', 'gloo')\n"
        "n = tl_dist.ranks_seen(group, 'cpu')\n"
        "t = torch.tensor([float(os.environ['RANK']) + 1.0], dtype=torch.float64)\n"
        "s = tl_dist.all_reduce_sum(t, group)\n"
        "if os.environ['RANK'] == '0':\n"
        "    print(json.dumps(dict(n=n, s=float(s), world=int(os.environ['WORLD_SIZE']))))\n"
        "torch.distributed.destroy_process_group()\n")
    driver = tmp_path / "driver.py"
    driver.write_text(
        "import sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from torchoptics_amd import dist as tl_dist\n"
        f"raise SystemExit(tl_dist.spawn_local_ranks({str(script)!r}, [], 2, timeout=240))\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cp = subprocess.run([sys.executable, str(driver)], capture_output=True, text=True, timeout=300, env=env)
    assert cp.returncode == 0, cp.stderr[-2000:]
    import json
    line = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1
    out = json.loads(line[0])
    assert out == dict(n=2, s=3.0, world=2)


def test_init_group_refuses_many_ranks_without_a_port(monkeypatch):
    from torchoptics_amd import dist as tl_dist
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.delenv("MASTER_PORT", raising=False)
    with pytest.raises(RuntimeError, match="MASTER_PORT"):
        tl_dist.init_group("cpu", "gloo")
