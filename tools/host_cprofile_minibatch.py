#!/usr/bin/env python3
"""cProfile of the eager minibatch step (examples/minibatch_loss.py: 256 lenses, ray aiming, penalty term): which Python
functions and torch calls the host spends its time in.  Development tool."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import minibatch_loss as mb          # noqa: E402
import torchoptics_amd as ta        # noqa: E402
from torchoptics_amd import ray_tracing as rt   # noqa: E402

dev = "cuda:0"
st, specs, leaves, n_seq = mb.build_batch(256, dev)
tracer = ta.RayTracer(mode="circular", n_rays=(8, 8), rel_fields=mb.FIELDS, wavelengths=mb.WAVELENGTHS, n_ray_aiming_iter=1,
                      default_device=dev)


def step():
    leaves["c"].grad = leaves["t"].grad = None
    lens = ta.Lens(st, leaves["c"], leaves["t"], leaves["nd"], leaves["v"])
    out = tracer.trace_rays(specs, lens, aggregate="sum")
    ld = rt.unsupervised_loss_batch(out, n_seq, 0.2)
    ld["loss_unsup"].sum().backward()


if "--loop" in sys.argv:
    # the reference's caller's loop (optical_loss.py:96-110): one lens per call, B = 1, full stacks
    n_l = 32
    singles = []
    for b_ in range(n_l):
        lv = {k: leaves[k].detach().reshape(256, -1)[b_].clone() for k in ("c", "t", "nd", "v")}
        lv["c"].requires_grad_(True), lv["t"].requires_grad_(True)
        singles.append((st[b_], specs[b_], lv))

    def step():          # noqa: F811
        for st1, sp1, lv in singles[:8]:
            lv["c"].grad = lv["t"].grad = None
            lens = ta.Lens(st1, lv["c"], lv["t"], lv["nd"], lv["v"])
            out = tracer.trace_rays(sp1, lens, aggregate=True)
            rt.unsupervised_loss(out, n_seq, 0.2)["loss_unsup"].backward()

for _ in range(30):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(300):
    step()
torch.cuda.synchronize()
print(f"eager step {1e6 * (time.perf_counter() - t0) / 300:.1f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
torch.cuda.synchronize()
ps = pstats.Stats(pr)
ps.sort_stats("tottime").print_stats(45)
