#!/usr/bin/env python3
"""Host-side wall time of the pieces of one eager bench step (tiny pupil: the GPU is never the bottleneck), so that
the per-call overhead of the autograd wrappers can be followed.  Development tool."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                      # noqa: E402
import torchoptics_amd as ta      # noqa: E402

args, meta, _ = bench.workload("cfg3", "cuda:0", 1, 0, 12)
leaves = [args[k] for k in bench.LEAF_NAMES if k in args]
N = 2000
t = {"zero": 0.0, "trace_skew": 0.0, "rms": 0.0, "backward": 0.0}
for it in range(N + 100):
    if it == 100:
        torch.cuda.synchronize()
        t = {k: 0.0 for k in t}
        t_all = time.perf_counter()
    a = time.perf_counter()
    for p in leaves:
        p.grad = None
    b = time.perf_counter()
    x, y, cx, cy, ok, back = ta.trace_skew(args["x"], args["y"], args["z"], args["cx"], args["cy"], args["c"], args["t"],
                                           args["mu"], args["mask"])
    c = time.perf_counter()
    rms = ta.compute_rms2d(x, y, ok)
    d = time.perf_counter()
    rms.backward()
    e = time.perf_counter()
    t["zero"] += b - a; t["trace_skew"] += c - b; t["rms"] += d - c; t["backward"] += e - d
torch.cuda.synchronize()
total = (time.perf_counter() - t_all) / N * 1e6
print({k: round(v / N * 1e6, 1) for k, v in t.items()}, "us per step; wall", round(total, 1), "us")
