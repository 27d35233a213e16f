#!/usr/bin/env python3
"""cProfile of the smallest eager step -- trace_skew + compute_rms2d + backward on a 1 M-ray Cooke fan -- i.e. the host chain
that bounds small workloads.  Development tool."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench          # noqa: E402
import torchoptics_amd as ta        # noqa: E402

job = bench.Job("cooke7", "cuda:0", 1, 0, None, 20, fields=(0.707,), wl=("d",))
for _ in range(50):
    job.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    job.step()
torch.cuda.synchronize()
print(f"eager step {1e6 * (time.perf_counter() - t0) / 500:.1f} us")
t0 = time.perf_counter()
for _ in range(500):
    job.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host time per step {1e6 * (t1 - t0) / 500:.1f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(500):
    job.step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
