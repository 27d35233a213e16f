"""cfg5-style optimisation loop on the GPU: Adam on c and t of the 20-row lens must reduce the
RMS spot monotonically enough (fwd + bwd through the HIP kernels every step)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_adam_reduces_rms_on_20_row_lens():
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import adam_loop
    out, losses = adam_loop.run(steps=40, lr=2e-4, log2_pupil=12, workload="zoom20")
    assert out["rows"] == 20 and out["fields"] == 5 and out["wavelengths"] == 3
    assert all(l == l and l > 0 for l in losses), "loss must stay finite"
    assert losses[-1] < 0.97 * losses[0], (losses[0], losses[-1])
    assert min(losses[20:]) <= min(losses[:20])


def test_trace_kernels_capture_in_a_hip_graph():
    """trace_skew + compute_rms2d + backward (both HIP kernels, the reductions and the tiny torch ops
    of the closed form) recorded into a HIP graph and replayed give the eager numbers: the C ABI does no
    allocation, synchronisation or host<->device copy on the launch path."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_golden
    import torchoptics_amd as ta
    g = load_golden("G2_cooke_16x16")
    names = ("in_x", "in_y", "in_z", "in_cx", "in_cy", "in_c", "in_t", "in_mu")
    ins = [torch.from_numpy(g[n]).cuda() for n in names]
    mask = torch.from_numpy(g["in_mask"]).cuda()
    for i in (5, 6, 7):
        ins[i].requires_grad_(True)

    def step():
        for i in (5, 6, 7):
            ins[i].grad = None
        out = ta.trace_skew(*ins, mask)
        loss = ta.compute_rms2d(out[0], out[1], out[4])
        loss.backward()
        return loss.detach()
    eager = step().item()
    eager_g = [ins[i].grad.clone() for i in (5, 6, 7)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for i in (5, 6, 7):
        ins[i].grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_loss = step()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert static_loss.item() == eager
    for i, ref in zip((5, 6, 7), eager_g):
        assert torch.equal(ins[i].grad, ref)


def _adam_child(*flags):
    import json
    import subprocess
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "adam_loop.py"), "--log2-pupil", "12", *flags],
                        capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert cp.returncode == 0, f"adam_loop {flags} exited with {cp.returncode}\n{cp.stderr[-3000:]}"
    return json.loads([ln for ln in cp.stdout.splitlines() if ln.startswith("{")][-1])


def test_whole_adam_step_replays_from_a_hip_graph():
    """The WHOLE optimisation step -- Lens assembly, dispersion, paraxial pupil position, both trace kernels, the
    RMS closed form, autograd over the host chain and Adam -- recorded once and replayed gives the eager loop's
    numbers.  (Round 1 crashed in hipStreamEndCapture here: the warm-up had run on the default stream and the
    leaves' AccumulateGrad nodes, kept alive by a Lens built outside the loop, belonged to that stream.)
    Child processes: the graph path warms up 3 steps before capturing, the eager loop 1, hence 10 vs 12 steps."""
    g = _adam_child("--graph", "--steps", "10")
    e = _adam_child("--capturable", "--steps", "12")
    assert g["hip_graph"] and not e["hip_graph"]
    assert g["loss_initial"] == e["loss_initial"]
    assert abs(g["loss_final"] - e["loss_final"]) <= 1e-6 * abs(e["loss_final"]), (g["loss_final"], e["loss_final"])
    assert g["loss_final"] != g["loss_initial"]          # the replays really stepped the parameters


def test_whole_adam_step_with_ray_aiming_replays_from_a_hip_graph():
    g = _adam_child("--graph", "--steps", "6", "--aim", "1")
    e = _adam_child("--capturable", "--steps", "8", "--aim", "1")
    assert abs(g["loss_final"] - e["loss_final"]) <= 1e-6 * abs(e["loss_final"]), (g["loss_final"], e["loss_final"])


def test_minibatch_of_lenses_batched_looped_and_replayed_from_a_graph():
    """examples/minibatch_loss.py: the reference caller's minibatch (aggregate + ray aiming, per-lens loss_unsup) as one
    batched launch, as the caller's one-lens-at-a-time loop, and replayed from a HIP graph: same losses, same gradients.
    (In a child process, like the other whole-step captures: a capture that goes wrong ends the process it runs in.)"""
    import json
    import subprocess
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "minibatch_loss.py"), "--lenses", "12", "--steps", "3",
                         "--loop-lenses", "4", "--graph"], capture_output=True, text=True, timeout=300)
    assert cp.returncode == 0, cp.stderr[-2000:]
    r = json.loads([ln for ln in cp.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["max_rel_loss_diff"] <= 1e-6 and r["grad_c_rel_diff"] <= 1e-5
    g = r["batched_hip_graph"]
    assert g["max_rel_loss_diff_vs_eager"] == 0.0 and g["grad_c_rel_diff_vs_eager"] == 0.0
    assert r["loss_mean"] > 0 and r["batched"]["lenses_per_s"] > r["one_lens_at_a_time"]["lenses_per_s"]
