#!/usr/bin/env python3
"""
The reference's real caller, as a minibatch: `Optical_Loss.optical_loss_unsupervised` (optical_loss.py:96-110) loops
over the samples of a minibatch and, for each, builds a `RaytracedOptics` (F = 8 fields, 8 x 8 circular pupil grid,
wavelengths 459 / 520 / 640 nm, one ray-aiming iteration) and calls `do_ray_tracing` -> loss_unsup = rms +
penalty_rate * sumQ: 1 536 rays per lens, one lens at a time.  Here the whole minibatch of B lenses is ONE padded
batch: one forward launch, one backward launch (tl_problem.B), per-lens losses from `unsupervised_loss_batch`.

    python examples/minibatch_loss.py --lenses 256 --steps 20          # batched, and the per-lens loop for comparison

Prints one JSON line: lenses/s and rays/s of the batched step and of the one-lens-at-a-time loop through the same
drop-in API, both including the whole host chain (dispersion, pupil position, ray aiming, loss, autograd).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FIELDS = tuple(float(v) for v in np.linspace(0, 1, 8))
WAVELENGTHS = (459., 520., 640.)
RINGS = 8
PENALTY_RATE = 0.2


def build_batch(n_lens, device, seed=0):
    """n_lens Cooke triplets, curvatures and thicknesses perturbed by +-2 % (what a generator network would emit)."""
    import yaml_free_lenses as L
    from torchoptics_amd import lens_modeling as lm
    a = L.PRESCRIPTIONS["cooke"]
    gen = torch.Generator().manual_seed(seed)
    S = len(a["c"])
    c = torch.tensor(a["c"]).repeat(n_lens, 1) * (1 + 0.02 * torch.randn(n_lens, S, generator=gen))
    t = torch.tensor(a["t"]).repeat(n_lens, 1) * (1 + 0.02 * torch.rand(n_lens, S, generator=gen))
    st = lm.Structure(stop_idx=np.array(a["stop_idx"] * n_lens), sequence=np.array(a["sequence"] * n_lens),
                      default_device=device)
    leaves = dict(c=c.reshape(-1).to(device).requires_grad_(True), t=t.reshape(-1).to(device).requires_grad_(True),
                  nd=torch.tensor(a["nd"] * n_lens, device=device), v=torch.tensor(a["v"] * n_lens, device=device))
    specs = lm.Specs(st, torch.full((n_lens,), L.EPD, device=device),
                     torch.full((n_lens,), float(np.deg2rad(L.HFOV_DEG)), device=device))
    return st, specs, leaves, len(a["sequence"][0])


def run(n_lens=256, steps=20, loop_lenses=32, aim=1, device="cuda:0", arith="strict", graph=False):
    import torchoptics_amd as ta
    from torchoptics_amd import ray_tracing as rt
    st, specs, leaves, n_seq = build_batch(n_lens, device)
    tracer = ta.RayTracer(mode="circular", n_rays=(RINGS, RINGS), rel_fields=FIELDS, wavelengths=WAVELENGTHS,
                          n_ray_aiming_iter=aim, default_device=device, arith=arith)
    rays_per_lens = len(FIELDS) * RINGS * RINGS * len(WAVELENGTHS)

    def batched_step(lv=leaves):
        lv["c"].grad = lv["t"].grad = None
        lens = ta.Lens(st, lv["c"], lv["t"], lv["nd"], lv["v"])
        out = tracer.trace_rays(specs, lens, aggregate=True)
        ld = rt.unsupervised_loss_batch(out, n_seq, PENALTY_RATE)
        ld["loss_unsup"].sum().backward()
        return ld

    S = st.mask.shape[1]
    singles = []
    for b in range(loop_lenses):
        lv = {k: leaves[k].detach().reshape(n_lens, -1)[b].clone() for k in ("c", "t", "nd", "v")}
        lv["c"].requires_grad_(True), lv["t"].requires_grad_(True)
        singles.append((st[b], specs[b], lv))

    def looped_step():
        losses = []
        for st1, sp1, lv in singles:
            lv["c"].grad = lv["t"].grad = None
            lens = ta.Lens(st1, lv["c"], lv["t"], lv["nd"], lv["v"])
            out = tracer.trace_rays(sp1, lens, aggregate=True)
            ld = rt.unsupervised_loss(out, n_seq, PENALTY_RATE)
            ld["loss_unsup"].backward()
            losses.append(ld["loss_unsup"].detach())
        return torch.stack(losses)

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            res = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n, res

    t_b, ld = timed(batched_step, steps)
    graph_res = None
    if graph:
        # the batched step recorded once into a HIP graph, from fresh leaves that only ever see the capture stream
        # (torchoptics_amd/graphs.py says why); a training loop copies the generator's new c, t into them and replays
        from torchoptics_amd import graphs
        gc_, gt_ = graphs.fresh_leaves(leaves["c"], leaves["t"])
        gl = dict(leaves, c=gc_, t=gt_)
        g, ld_g = graphs.capture_step(lambda: batched_step(gl), device)
        new_c, new_t = leaves["c"].detach().clone(), leaves["t"].detach().clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            with torch.no_grad():
                gl["c"].copy_(new_c), gl["t"].copy_(new_t)              # "the generator's output of this step"
            g.replay()
        torch.cuda.synchronize()
        t_g = (time.perf_counter() - t0) / steps
        graph_res = dict(ms_per_step=t_g * 1e3, lenses_per_s=n_lens / t_g, M_rays_per_s=n_lens * rays_per_lens / t_g / 1e6,
                         max_rel_loss_diff_vs_eager=float(((ld_g["loss_unsup"].detach() - ld["loss_unsup"].detach()).abs()
                                                           / ld["loss_unsup"].detach().abs()).max()),
                         grad_c_rel_diff_vs_eager=float((gl["c"].grad - leaves["c"].grad).norm() / leaves["c"].grad.norm()))
    t_l, l_loop = timed(looped_step, max(1, steps // 4))
    # the batch and the loop compute the same per-lens losses and gradients
    lb = ld["loss_unsup"].detach()[:loop_lenses]
    g_b = leaves["c"].grad.reshape(n_lens, S)[:loop_lenses]
    g_l = torch.stack([lv["c"].grad for _, _, lv in singles])
    return dict(
        workload=f"{n_lens} perturbed Cooke triplets x F=8 x {RINGS}x{RINGS} pupil grid x W=3 = {rays_per_lens} rays per lens, "
                 f"aggregate=True, ray aiming {aim}, loss_unsup = rms + {PENALTY_RATE} sumQ per lens, fwd+bwd, whole host chain",
        arith_mode=arith,
        batched=dict(ms_per_step=t_b * 1e3, lenses_per_s=n_lens / t_b, M_rays_per_s=n_lens * rays_per_lens / t_b / 1e6,
                     launches="1 forward + 1 backward kernel for the whole minibatch"),
        one_lens_at_a_time=dict(lenses=loop_lenses, ms_per_lens=t_l / loop_lenses * 1e3, lenses_per_s=loop_lenses / t_l,
                                M_rays_per_s=loop_lenses * rays_per_lens / t_l / 1e6,
                                note="the reference's caller's loop (optical_loss.py:96-110) through the same API, B = 1 per call"),
        batched_hip_graph=graph_res,
        speedup=(n_lens / t_b) / (loop_lenses / t_l),
        max_rel_loss_diff=float(((lb - l_loop).abs() / l_loop.abs()).max()),
        grad_c_rel_diff=float((g_b - g_l).norm() / g_l.norm()),
        loss_mean=float(ld["loss_unsup"].detach().mean()), rms_mean=float(ld["rms"].detach().mean()),
        penalty_mean=float(ld["penalty"].detach().mean()))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--lenses", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--loop-lenses", type=int, default=32)
    ap.add_argument("--aim", type=int, default=1)
    ap.add_argument("--mode", default="strict")
    ap.add_argument("--graph", action="store_true", help="also replay the batched step from a HIP graph")
    a = ap.parse_args()
    print(json.dumps(run(a.lenses, a.steps, a.loop_lenses, a.aim, arith=a.mode, graph=a.graph)))
