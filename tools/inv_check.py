#!/usr/bin/env python3
"""Compare the two backward algorithms (checkpoint vs inverse) on a workload: gradients and time."""
import os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import torchoptics_amd as ta
from torchoptics_amd import ops
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
mode = sys.argv[2] if len(sys.argv) > 2 else "strict"
lp = int(sys.argv[3]) if len(sys.argv) > 3 else None
args, meta, _ = bench.workload(wl, "cuda:0", 1, 0, lp)
names = [k for k in bench.LEAF_NAMES if k in args]
res = {}
for algo in ("checkpoint", "inverse"):
    ops.set_backward_algorithm(algo)
    times = []
    for it in range(8):
        for k in names: args[k].grad = None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x, y, cx, cy, ok, back = ta.trace_skew(args["x"], args["y"], args["z"], args["cx"], args["cy"], args["c"], args["t"], args["mu"], args["mask"], mode=mode)
        loss = ta.compute_rms2d(x, y, ok)
        e0.record(); loss.backward(); e1.record(); torch.cuda.synchronize()
        if it > 1: times.append(e0.elapsed_time(e1))
    res[algo] = (loss.item(), {k: args[k].grad.clone() for k in names}, statistics.median(times))
a, b = res["checkpoint"], res["inverse"]
print(f"{wl} {mode}: loss {a[0]:.9g} / {b[0]:.9g}; backward (incl. tiny ops) checkpoint {a[2]:.4f} ms, inverse {b[2]:.4f} ms")
for k in names:
    d = (a[1][k].double() - b[1][k].double()).norm() / a[1][k].double().norm().clamp_min(1e-300)
    print(f"   d/d{k}: inverse vs checkpoint rel {d.item():.2e}")
