#!/usr/bin/env python3
"""
cfg5 of BASELINE.json: an Adam optimisation loop on the lens parameters (curvatures and thicknesses)
of the 20-row synthetic zoom, 5 fields x 3 wavelengths, every step = forward + RMS spot + backward
through the HIP kernels; the pupil is sharded over the GPUs of one node.

    python examples/adam_loop.py --steps 100                       # 1 GPU
    python examples/adam_loop.py --steps 100 --gpus 8              # starts its own 8 ranks (torch.distributed.run)
    python -m torch.distributed.run --nproc-per-node 8 examples/adam_loop.py --steps 100

The optimiser state lives on the device and nothing in the loop synchronises with the host except
the final report; per step and rank: 1 forward kernel, 1 all-reduce of [F,10] doubles, 1 backward
kernel, 1 all-reduce of the packed leaf gradients, Adam on < 100 scalars.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(steps=100, lr=2e-4, log2_pupil=20, workload="zoom20", device="cuda:0", group=None, rank=0, world=1,
        arith="strict", verbose=False, graph=False, aim=0, capturable=None):
    import torchoptics_amd as ta
    from torchoptics_amd import dist as tl_dist, prescriptions as P, ray_tracing as rt
    lens_fn = {"zoom20": P.zoom20, "double_gauss": P.double_gauss}[workload]
    # The prescription builder returns a Lens whose padded c/t are VIEWS of the leaves: a piece of autograd
    # graph (and with it the leaves' AccumulateGrad nodes, which remember the stream they were created on)
    # that would stay alive across steps.  Keep only the structure, the specs and the bare leaves.
    lens0, specs, leaves = lens_fn(device)
    structure, n_rows = lens0.structure, int(lens0.c.shape[1])
    del lens0
    fields = tuple(np.linspace(0, 1, 5)) if workload == "zoom20" else (0., 0.707, 1.)
    wl = ("C", "d", "F")
    p_local = 1 << log2_pupil
    n_r = 1 << (log2_pupil // 2)
    n_theta = (p_local // n_r) * world
    tracer = ta.RayTracer(mode="circular", n_rays=(n_r, n_theta), rel_fields=fields, wavelengths=wl,
                          default_device=device, arith=arith, n_ray_aiming_iter=aim)
    xy = rt.circle_index_range(n_r, n_theta, rank * p_local, (rank + 1) * p_local, device)
    params = [leaves["c"], leaves["t"]]
    nd, v = leaves["nd"].detach(), leaves["v"].detach()
    opt = torch.optim.Adam(params, lr=lr, capturable=graph if capturable is None else capturable)
    n_per_field = p_local * world * len(wl)
    history = []

    def one_step():
        opt.zero_grad(set_to_none=True)
        lens = ta.Lens(structure, leaves["c"], leaves["t"], nd, v)
        x, y, cx, cy, ok, back = tracer.trace_rays(specs, lens, xy=xy)
        loss = rt.compute_rms2d(x, y, ok, group=group, n_per_field=n_per_field)
        loss.backward()
        if group is not None:
            tl_dist.all_reduce_grads(params, group)
        opt.step()
        return loss.detach()

    if graph:
        # Whole-step HIP graph: forward kernel, moments all-reduce, closed form, backward kernel, the host chain's
        # autograd (dispersion, pupil position), gradient all-reduce and Adam are recorded once and replayed.
        # Every autograd node the captured backward touches must live on the CAPTURE stream: the engine runs a
        # leaf's AccumulateGrad on the stream that node was created on, and a node left over from a default-
        # stream warm-up makes the engine synchronise the capturing stream with the legacy default stream --
        # which hipStreamEndCapture on ROCm answers with a segfault (round-1 gpurun_out/graph_dbg.log).  So: warm
        # up ON the capture stream, keep no graph alive between steps, capture on that same stream.
        cap = torch.cuda.Stream(device)
        cap.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(cap):
            for _ in range(3):
                history.append(one_step())     # warm-up: allocations, Adam state, workspace of this stream
        torch.cuda.current_stream(device).wait_stream(cap)
        torch.cuda.synchronize()
        history = history[:1]                  # the initial loss; the other warm-up steps are not reported
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            static_loss = one_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            g.replay()
            history.append(static_loss.clone())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    else:
        history.append(one_step())          # warm-up (allocations), also the initial loss
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            history.append(one_step())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    losses = torch.stack(history).cpu().tolist()
    rays = len(fields) * len(wl) * p_local * world
    out = dict(workload=workload, rows=n_rows, fields=len(fields), wavelengths=len(wl), rays_per_step=rays,
               n_gpus=world, steps=steps, steps_per_s=steps / dt, M_rays_per_s=rays * steps / dt / 1e6,
               loss_initial=losses[0], loss_final=losses[-1], arith_mode=arith, hip_graph=bool(graph))
    if verbose and rank == 0:
        print(json.dumps(out))
    return out, losses


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--log2-pupil", type=int, default=20)
    ap.add_argument("--workload", default="zoom20", choices=["zoom20", "double_gauss"])
    ap.add_argument("--mode", default="strict", choices=["strict", "fast"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--graph", action="store_true", help="record the whole optimisation step into a HIP graph and replay it")
    ap.add_argument("--capturable", action="store_true", help="eager loop with Adam(capturable=True): the arithmetic of the graph path")
    ap.add_argument("--aim", type=int, default=0, help="n_ray_aiming_iter (the reference's real caller uses 1)")
    ap.add_argument("--force-dist", action="store_true", help="initialise a 1-rank nccl (RCCL) group even when WORLD_SIZE=1")
    ap.add_argument("--gpus", type=int, default=1, help="ranks to start (one per GPU) when not already under a launcher")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # start the ranks ourselves, as a child process tree, before this process touches the GPU
        if a.backend == "nccl" and a.gpus > torch.cuda.device_count():
            raise SystemExit(f"--gpus {a.gpus} but this node shows {torch.cuda.device_count()} GPU(s)")
        from torchoptics_amd import dist as tl_dist
        raise SystemExit(tl_dist.spawn_local_ranks(os.path.abspath(__file__), sys.argv[1:], a.gpus))
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(local)
    from torchoptics_amd import dist as tl_dist
    group = tl_dist.init_group(f"cuda:{local}", a.backend, force=a.force_dist)
    run(a.steps, a.lr, a.log2_pupil, a.workload, f"cuda:{local}", group, rank, world, a.mode, verbose=True, graph=a.graph, aim=a.aim,
        capturable=True if a.capturable else None)
    if group is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
