#!/usr/bin/env python3
"""cProfile of the host side of one Adam-loop step (tiny pupil: the GPU is never the bottleneck).
Development tool: python tools/host_profile_adam.py [--steps 200]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    import torchoptics_amd as ta
    from torchoptics_amd import prescriptions as P, ray_tracing as rt
    device = "cuda:0"
    lens0, specs, leaves = P.zoom20(device)
    tracer = ta.RayTracer(mode="circular", n_rays=(32, 32), rel_fields=tuple(np.linspace(0, 1, 5)),
                          wavelengths=("C", "d", "F"), default_device=device)
    xy = rt.circle_index_range(32, 32, 0, 1024, device)
    params = [leaves["c"], leaves["t"]]
    opt = torch.optim.Adam(params, lr=2e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        lens = ta.Lens(lens0.structure, leaves["c"], leaves["t"], leaves["nd"].detach(), leaves["v"].detach())
        x, y, cx, cy, ok, back = tracer.trace_rays(specs, lens, xy=xy)
        rt.compute_rms2d(x, y, ok).backward()
        opt.step()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) / a.steps * 1e3:.3f} ms per step (host-bound)")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
