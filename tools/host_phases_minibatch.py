#!/usr/bin/env python3
"""Host wall time per phase of one eager minibatch step (examples/minibatch_loss.py; 256 lenses, ray aiming, penalty
term): no device synchronisation inside the loop, so this is the time the host spends ISSUING each phase, and kernel
launches per phase from the torch profiler.  Development tool."""
import os
import sys
import time

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
T = {}


def lap(key, t0):
    t1 = time.perf_counter()
    T[key] = T.get(key, 0.0) + (t1 - t0)
    return t1


def step():
    t = time.perf_counter()
    leaves["c"].grad = leaves["t"].grad = None
    lens = ta.Lens(st, leaves["c"], leaves["t"], leaves["nd"], leaves["v"])
    t = lap("Lens()", t)
    n = lens.get_refractive_indices(tracer.wavelengths)
    t = lap("dispersion", t)
    z = rt.compute_pupil_position(lens, tracer.arith)
    t = lap("pupil position", t)
    aim = tracer.ray_aiming(specs, lens.detach(), True)
    t = lap("ray aiming (trace to stop + Jacobian)", t)
    a = tracer.assemble(specs, lens)
    t = lap("assemble (incl. the three above again)", t)
    out = rt.trace_skew(a['x'], a['y'], a['z'], a['cx'], a['cy'], a['c'], a['t'], a['mu'], a['mask'], True, True)
    t = lap("trace_skew", t)
    ld = rt.unsupervised_loss_batch(out, n_seq, 0.2)
    loss = ld["loss_unsup"].sum()
    t = lap("loss", t)
    loss.backward()
    t = lap("backward", t)


for _ in range(20):
    step()
torch.cuda.synchronize()
T.clear()
N = 200
t_all = time.perf_counter()
for _ in range(N):
    step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t_all) / N * 1e6
for k, v in T.items():
    print(f"{k:45s} {v / N * 1e6:8.1f} us")
print(f"{'sum of phases':45s} {sum(T.values()) / N * 1e6:8.1f} us   (wall incl. GPU drain {wall:.1f} us; note: this step does the "
      "dispersion / pupil position / aiming twice, once stand-alone and once inside assemble)")
