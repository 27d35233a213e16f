#!/usr/bin/env python3
"""Diagnostic: where does the gradient difference between the HIP path and CPU autograd come from?
Compares, on a sample of the bench workload: oracle fp32 (torch.sqrt), oracle fp32 (IEEE sqrt),
oracle fp64, HIP strict, HIP fast; prints norm-relative differences per parameter group."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import torchoptics_amd as ta
from oracle import trace_oracle as orc

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg3"); ap.add_argument("--log2", type=int, default=20)
a = ap.parse_args()
args, meta, _ = bench.workload(a.workload, "cuda:0", 1, 0, a.log2)
names = ("z", "cy", "c", "t", "mu")

def rel(x, y):
    return float(((x.double() - y.double()).norm() / y.double().norm().clamp_min(1e-300)).item())

def cpu(dt, ieee):
    src = {k: (v.detach().cpu().to(dt) if v.is_floating_point() else v.detach().cpu()) for k, v in args.items()}
    lv = [src[k].requires_grad_(True) for k in names]
    x, y, cx, cy, ok, back = orc.trace_skew(src["x"], src["y"], src["z"], src["cx"], src["cy"], src["c"], src["t"], src["mu"], src["mask"], ieee_sqrt=ieee)
    loss = orc.compute_rms2d(x, y, ok); loss.backward()
    return loss.item(), [q.grad for q in lv]

def gpu(mode):
    gl = [args[k].detach().clone().requires_grad_(True) for k in names]
    ga = dict(args); ga.update(dict(zip(names, gl)))
    x, y, cx, cy, ok, back = ta.trace_skew(ga["x"], ga["y"], ga["z"], ga["cx"], ga["cy"], ga["c"], ga["t"], ga["mu"], ga["mask"], mode=mode)
    loss = ta.compute_rms2d(x, y, ok); loss.backward()
    return loss.item(), [q.grad.cpu() for q in gl]

res = {"cpu32_mkl": cpu(torch.float32, False), "cpu32_ieee": cpu(torch.float32, True), "cpu64": cpu(torch.float64, False),
       "hip_strict": gpu("strict"), "hip_fast": gpu("fast")}
print("rms:", {k: f"{v[0]:.10g}" for k, v in res.items()})
truth = res["cpu64"][1]
for k, (_, g) in res.items():
    print(f"{k:12s} vs fp64 : " + "  ".join(f"{n}={rel(a_, b_):.2e}" for n, a_, b_ in zip(names, g, truth)))
for other in ("cpu32_mkl", "cpu32_ieee"):
    for k in ("hip_strict", "hip_fast"):
        print(f"{k:12s} vs {other:10s}: " + "  ".join(f"{n}={rel(a_, b_):.2e}" for n, a_, b_ in zip(names, res[k][1], res[other][1])))
print("cpu32_mkl    vs cpu32_ieee: " + "  ".join(f"{n}={rel(a_, b_):.2e}" for n, a_, b_ in zip(names, res["cpu32_mkl"][1], res["cpu32_ieee"][1])))
