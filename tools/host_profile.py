#!/usr/bin/env python3
"""cProfile of the host side of one bench step (tiny pupil, so the GPU is never the bottleneck).
Development tool: python tools/host_profile.py [--workload cfg2] [--steps 300]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--steps", type=int, default=300)
    a = ap.parse_args()
    import bench
    import torchoptics_amd as ta
    args, meta, _ = bench.workload(a.workload, "cuda:0", 1, 0, 12)
    leaves = [args[k] for k in bench.LEAF_NAMES if k in args]
    asph = {k: args[k] for k in ("kappa", "poly") if k in args}

    def step():
        for p in leaves:
            p.grad = None
        x, y, cx, cy, ok, back = ta.trace_skew(args["x"], args["y"], args["z"], args["cx"], args["cy"], args["c"],
                                               args["t"], args["mu"], args["mask"], **asph)
        ta.compute_rms2d(x, y, ok).backward()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) / a.steps * 1e3:.3f} ms per step (host-bound, P={meta['P_local']})")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)


if __name__ == "__main__":
    main()
