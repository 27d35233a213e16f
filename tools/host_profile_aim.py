#!/usr/bin/env python3
"""Host cost of trace_rays with and without one ray-aiming iteration (tiny pupil: GPU never the bottleneck)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import cProfile
    import pstats
    import torchoptics_amd as ta
    from torchoptics_amd import prescriptions as P, ray_tracing as rt
    device = "cuda:0"
    lens0, specs, leaves = P.double_gauss(device)
    for aim in (0, 1):
        tracer = ta.RayTracer(mode="circular", n_rays=(32, 32), rel_fields=(0., 0.707, 1.), wavelengths=("C", "d", "F"),
                              n_ray_aiming_iter=aim, default_device=device)

        def step():
            for q in (leaves["c"], leaves["t"]):
                q.grad = None
            lens = ta.Lens(lens0.structure, leaves["c"], leaves["t"], leaves["nd"].detach(), leaves["v"].detach())
            x, y, cx, cy, ok, back = tracer.trace_rays(specs, lens)
            rt.compute_rms2d(x, y, ok).backward()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            step()
        torch.cuda.synchronize()
        print(f"n_ray_aiming_iter={aim}: {(time.perf_counter() - t0) / 100 * 1e3:.3f} ms per step (host-bound)")
        if aim:
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(100):
                step()
            torch.cuda.synchronize()
            pr.disable()
            pstats.Stats(pr).sort_stats("cumulative").print_stats(30)


if __name__ == "__main__":
    main()
