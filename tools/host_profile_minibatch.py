#!/usr/bin/env python3
"""Where the host time of one batched minibatch step goes (examples/minibatch_loss.py): torch profiler table of CPU-side
op time and kernel launch counts for B = 256 lenses.  Development tool."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

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
    out = tracer.trace_rays(specs, lens, aggregate=True)
    rt.unsupervised_loss_batch(out, n_seq, 0.2)["loss_unsup"].sum().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=35, max_name_column_width=60))
n_launch = sum(e.count for e in prof.key_averages() if e.key in ("hipLaunchKernel", "hipExtModuleLaunchKernel", "hipModuleLaunchKernel"))
print("kernel launches per step:", n_launch / 5)
