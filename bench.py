#!/usr/bin/env python3
"""
bench.py -- rays/s of the sequential ray-trace hot path, forward + backward, on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg3a|cfg2|cfg5] [--mode strict|fast]

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
(one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

A "step" is one pass of the hot path over one resident batch of rays: trace_skew (fused forward
kernel, per-ray outputs materialised as the reference API returns them, spot moments fused) ->
compute_rms2d (closed form on the moments; one tiny all-reduce when N > 1) -> backward (recompute
+ adjoint kernel) -> gradients of the loss w.r.t. the trace parameters (c, t, mu, z, cy) [-> one
tiny all-reduce when N > 1].  The pupil coordinates are resident in HBM before the timed region.

Workloads:
  cfg3 (default): synthesized double Gauss, 11 rows (10 refracting surfaces + stop), F=1 field
        (0.707), W=1 ('d'), circular pupil grid 4096 x 4096 = 2^24 rays PER GPU (weak scaling; at
        N=8 this is cfg4's 2^27 rays).  This is the config BASELINE.json's metric ("M rays/s
        through 10-surface lens; fwd+bwd") is quoted on.
  cfg3a: the same double Gauss with 2 aspheric rows (conic + a4, a6; Newton intersection) -- BASELINE
        configs[2] as written; an extension beyond the reference (parity unpinned by it).
  cfg2: Cooke triplet (7 rows), 1024 x 1024 pupil, 3 fields, W=1.
  cfg5: 20-row synthetic zoom, 5 fields x 3 wavelengths, 1024 x 1024 pupil.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

LEAF_NAMES = ("z", "cy", "c", "t", "mu", "kappa", "poly")   # differentiable arguments of trace_skew
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: peak FP32 vector (packed FMA)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=["cfg3", "cfg3a", "cfg2", "cfg5"])
    ap.add_argument("--mode", default=os.environ.get("TORCHOPTICS_AMD_MODE", "strict"), choices=["strict", "fast"])
    ap.add_argument("--log2-pupil", type=int, default=None, help="override log2 of pupil points per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse several ranks on one GPU)")
    ap.add_argument("--no-other-mode", action="store_true", help="skip the secondary measurement of the other arithmetic mode")
    ap.add_argument("--no-also", action="store_true", help="skip the extra workloads reported next to the main one")
    ap.add_argument("--graph", action="store_true",
                    help="record one step (both kernels + reductions + closed form + autograd bookkeeping) into a HIP "
                         "graph and time replays of it instead of eager steps")
    ap.add_argument("--no-graph-child", action="store_true", help="skip the secondary HIP-graph measurement (child process)")
    ap.add_argument("--cpu-log2-rays", type=int, default=20, help="log2 of the CPU-baseline sample (rays)")
    return ap.parse_args()


def workload(name, device, world, rank, log2_pupil):
    """Returns dict(lens args as leaves, pupil slice, meta)."""
    import torchoptics_amd as ta
    from torchoptics_amd import prescriptions as P, ray_tracing as rt
    if name in ("cfg3", "cfg3a"):
        lens, specs, leaves = P.double_gauss(device, aspheres=(name == "cfg3a"))
        fields, wl, lp = (0.707,), ("d",), 24
    elif name == "cfg2":
        import yaml_free_lenses as L
        lens, specs, leaves = L.build("cooke", device)
        fields, wl, lp = (0., 0.707, 1.), ("d",), 20
    else:
        lens, specs, leaves = P.zoom20(device)
        fields, wl, lp = tuple(np.linspace(0, 1, 5)), ("C", "d", "F"), 20
    lp = log2_pupil if log2_pupil is not None else lp
    p_local = 1 << lp
    n_r = 1 << (lp // 2)
    n_theta_total = (p_local // n_r) * world            # weak scaling: the grid grows with N
    tr = ta.RayTracer(mode="circular", n_rays=(n_r, n_theta_total), rel_fields=fields, wavelengths=wl,
                      default_device=device)
    xy = rt.circle_index_range(n_r, n_theta_total, rank * p_local, (rank + 1) * p_local, device)
    with torch.no_grad():
        a = tr.assemble(specs, lens, xy=xy)
    args = {k: v.detach().clone() for k, v in a.items() if k != "n_index"}
    for k in LEAF_NAMES:
        if k in args:
            args[k].requires_grad_(True)
    meta = dict(F=len(fields), W=len(wl), S=lens.c.shape[1], P_local=p_local, P_total=p_local * world,
                lens=name, fields=list(map(float, fields)), wavelengths=list(wl))
    return args, meta, (tr, specs, lens, leaves, xy)


def pmc_traffic(workload_name, mode, kernel, meta):
    """HBM bytes per launch measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes,
    gfx950 FETCH correction applied) for this workload at its default size; None if not profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            d = json.load(f)
        if meta["P_local"] != (1 << 24):
            return None
        return d[workload_name][mode][kernel]["hbm_bytes"]
    except (OSError, KeyError, ValueError):
        return None


def flops_per_ray(S):
    """fp32 arithmetic operations per ray as written in csrc/tl_kernels.inc (mul, add/sub, sqrt, div,
    rcp each count 1; compares, selects, negations not counted; counted by hand):
    step_fwd 54 per surface (+5 image plane); step_bwd 49 recompute + 99 adjoint per surface; the
    backward kernel runs step_fwd once more to reach the image plane (+30 seeds/entrance)."""
    fwd = 54 * S + 5
    bwd = fwd + (49 + 99) * S + 30          # checkpoint kernel: forward sweep + recompute + adjoint
    bwd_inv = (51 + 17 + 99) * S + 40       # walk-back kernel: inverse refraction + intersection, partial recompute, adjoint
    return fwd, bwd, bwd_inv


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    assert torch.cuda.is_available(), "bench.py needs an AMD GPU"
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    group = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group("gloo")
        group = dist.group.WORLD

    import torchoptics_amd as ta
    from torchoptics_amd import _lib, ops, dist as tl_dist
    _lib.lib()      # fail loudly when the HIP library is missing
    ops.set_default_mode(a.mode)

    args, meta, extra = workload(a.workload, device, world, rank, a.log2_pupil)
    leaves = [args[k] for k in LEAF_NAMES if k in args]
    asph = {k: args[k] for k in ("kappa", "poly") if k in args}
    n_per_field_total = meta["P_total"] * meta["W"]

    def step():
        for p in leaves:
            p.grad = None
        x, y, cx, cy, ok, back = ta.trace_skew(args["x"], args["y"], args["z"], args["cx"], args["cy"], args["c"],
                                               args["t"], args["mu"], args["mask"], **asph)
        rms = ta.compute_rms2d(x, y, ok, group=group, n_per_field=n_per_field_total)
        rms.backward()
        if group is not None:
            tl_dist.all_reduce_grads(leaves, group)
        return rms

    def sync():
        if group is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(mode):
        """W untimed warm-up steps, then exactly K steps between barrier+synchronize; max over ranks."""
        ops.set_default_mode(mode)
        if a.graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(a.warmup, 3)):
                    r = step()
            torch.cuda.current_stream().wait_stream(side)
            for p_ in leaves:
                p_.grad = None
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                r = step()
            for _ in range(a.warmup):
                g.replay()
            sync()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                g.replay()
            sync()
            el, km = time.perf_counter() - t0, {}
        else:
            for _ in range(a.warmup):
                r = step()
            sync()
            ops.enable_timing(True)
            t0 = time.perf_counter()
            for _ in range(a.steps):
                r = step()
            sync()
            el = time.perf_counter() - t0
            km = ops.timing_ms()
            ops.enable_timing(False)
        if group is not None:
            tmax = torch.tensor([el], dtype=torch.float64, device="cpu" if a.backend == "gloo" else device)
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
            el = tmax.item()
        return el, km, r

    dt, kern_ms, rms = timed(a.mode)

    rays_local = meta["F"] * meta["W"] * meta["P_local"]
    rays_total = rays_local * world
    ms_per_step = dt / a.steps * 1e3
    value = rays_total * a.steps / dt / 1e6

    # ---- roofline of the dominant kernel (trace_bwd_kernel), per launch, from live event timing
    fw = meta["F"] * meta["W"]
    from torchoptics_amd import ops as _ops
    inv = _ops.get_backward_algorithm() == "inverse"       # aspheric rows are walked back too
    # algorithmic bytes per ray (DESIGN.md "bytes per unit"): forward writes x,y,cx,cy,ok,back and reads x_in,y_in;
    # the walk-back backward reads x_in,y_in and the forward's x,y,cx,cy,ok; the checkpoint backward only x_in,y_in
    b_fwd, b_bwd = 18.0 + 8.0 / fw, (17.0 if inv else 0.0) + 8.0 / fw
    f_fwd, f_ck, f_inv = flops_per_ray(meta["S"])     # counted for spherical rows; aspheric rows cost more (not counted)
    f_bwd = f_inv if inv else f_ck
    bwd_kernel_name = "trace_bwd_inv_kernel" if inv else "trace_bwd_kernel"
    kernels = {}
    for key, bpr, fpr in (("fwd", b_fwd, f_fwd), ("bwd", b_bwd, f_bwd)):
        ms = kern_ms.get(key)
        if ms:
            kernels[key] = dict(ms=ms, rays_per_s=rays_local / ms * 1e3, hbm_GBs=rays_local * bpr / ms / 1e6,
                                valu_TFLOPs=rays_local * fpr / ms / 1e9, bytes_per_ray=bpr, flops_per_ray=fpr)
    dom = kernels.get("bwd")
    roofline = None
    if dom:
        roofline = dict(kernel=bwd_kernel_name, bound="hbm", achieved=dom["hbm_GBs"], peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=dom["hbm_GBs"] / HBM_PEAK_GBS, traffic=pmc_traffic(a.workload, a.mode, "bwd", meta),
                        launch_ms=dom["ms"], algorithmic_bytes_per_launch=rays_local * b_bwd,
                        note="per-ray FMA kernel: the binding limit is the FP32 vector ALU, see roofline_valu; "
                             "traffic = HBM bytes per launch from the committed rocprofv3 PMC passes "
                             "(profiles/r01_pmc_traffic.json), null for workloads not profiled")
    roofline_valu = None
    if dom:
        roofline_valu = dict(kernel=bwd_kernel_name, bound="valu_fp32", achieved=dom["valu_TFLOPs"],
                             peak=VALU_PEAK_TFLOPS, unit="TFLOP/s", frac=dom["valu_TFLOPs"] / VALU_PEAK_TFLOPS)
    # whole-step algorithmic HBM rate (fwd + bwd bytes at the API boundary, SURVEY 8d headline)
    step_bytes = rays_total * (b_fwd + b_bwd)

    other = None
    if not a.no_other_mode:
        om = "fast" if a.mode == "strict" else "strict"
        odt, okm, orms = timed(om)
        other = dict(arith_mode=om, value=rays_total * a.steps / odt / 1e6, unit="M rays/s", ms_per_step=odt / a.steps * 1e3,
                     fwd_kernel_ms=okm.get("fwd"), bwd_kernel_ms=okm.get("bwd"), rms=float(orms.item()))
        ops.set_default_mode(a.mode)

    # secondary workloads, same protocol (W warm-up + K timed steps), reported under "also"
    also = {}
    if not a.no_also and a.workload == "cfg3" and a.log2_pupil is None:
        main_state = (args, meta, leaves, asph, n_per_field_total)
        for wname in ("cfg3a", "cfg2", "cfg5"):
            args, meta2, _ = workload(wname, device, world, rank, None)
            leaves = [args[k] for k in LEAF_NAMES if k in args]
            asph = {k: args[k] for k in ("kappa", "poly") if k in args}
            n_per_field_total = meta2["P_total"] * meta2["W"]
            adt, akm, arms = timed(a.mode)
            nr = meta2["F"] * meta2["W"] * meta2["P_local"] * world
            also[wname] = dict(value=nr * a.steps / adt / 1e6, unit="M rays/s", ms_per_step=adt / a.steps * 1e3,
                               fwd_kernel_ms=akm.get("fwd"), bwd_kernel_ms=akm.get("bwd"), rays=nr, rows=meta2["S"],
                               F=meta2["F"], W=meta2["W"], rms=float(arms.item()), arith_mode=a.mode)
            if akm.get("fwd") and akm.get("bwd"):
                # small workloads are bound by the eager Python/autograd chain (~0.25-0.4 ms of host time per
                # step, box dependent), not by the GPU: also quote what the two trace kernels alone sustain
                also[wname]["trace_kernels_only_value"] = nr / (akm["fwd"] + akm["bwd"]) / 1e3
        args, meta, leaves, asph, n_per_field_total = main_state

    # the same step replayed from a HIP graph, measured in a CHILD process (a capture failure of the
    # PyTorch/ROCm stack must not cost the main line); N = 1 only
    hip_graph = None
    if world == 1 and not a.graph and not a.no_graph_child:
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--graph", "--steps", str(a.steps), "--warmup", str(a.warmup),
               "--workload", a.workload, "--mode", a.mode, "--no-cpu-baseline", "--no-other-mode", "--no-also"]
        if a.log2_pupil is not None:
            cmd += ["--log2-pupil", str(a.log2_pupil)]
        try:
            cp = subprocess.run(cmd, capture_output=True, text=True, timeout=180)
            line = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
            if cp.returncode == 0 and line:
                child = json.loads(line[-1])
                hip_graph = dict(value=child["value"], unit="M rays/s", ms_per_step=child["ms_per_step"],
                                 note="the identical step recorded once into a HIP graph and replayed")
            else:
                hip_graph = dict(error=f"child exited with {cp.returncode}")
        except Exception as e:       # timeout or launch failure
            hip_graph = dict(error=repr(e))

    cpu_baseline, grad_check = None, None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu_baseline, grad_check = cpu_leg(args, meta, a.cpu_log2_rays, a.mode)

    if rank == 0:
        out = {
            "metric": "M rays/s through 10-surface lens; fwd+bwd; grad rel-err vs PyTorch autograd",
            "value": value, "unit": "M rays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.workload}: {meta['lens']} S={meta['S']} rows, F={meta['F']} W={meta['W']} "
                                   f"P={meta['P_local']} pupil points per GPU ({rays_local} rays/GPU, {rays_total} total), "
                                   f"circular grid, loss=compute_rms2d, fwd+bwd",
                       "arith_mode": a.mode, "backward_algorithm": "walk-back from the forward outputs (checkpoint kernel "
                       "as on-device fallback for ill-conditioned fans)" if inv else "checkpoint",
                       "parallelism": f"pupil-sharded dp{world}", "rms": float(rms.item()),
                       "hip_graph": bool(a.graph)},
            "roofline": roofline, "roofline_valu": roofline_valu, "kernels": kernels,
            "step_hbm_GBs": step_bytes / (dt / a.steps) / 1e9,
            "grad_rel_err_vs_pytorch_autograd": grad_check,
            "other_mode": other,
            "hip_graph_replay": hip_graph,
            "also": also,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out))
    if group is not None:
        torch.distributed.destroy_process_group()


def cpu_leg(args, meta, log2_rays, mode):
    """The oracle (CPU restatement of the reference, eager PyTorch fp32 + autograd) timed on this
    box's host cores on a bounded sample of the same workload: the first 2^log2_rays/(F*W) pupil
    points.  kind='port': the reference itself cannot travel to the GPU box.  The same sample is then
    traced by the HIP path and the gradients are compared (norm-relative, per parameter group)."""
    import torchoptics_amd as ta
    from oracle import trace_oracle as orc
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("TL_CPU_THREADS", "32"))))
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: oracle on {cores} threads ...", file=sys.stderr, flush=True)
    fw = meta["F"] * meta["W"]
    p = max(1, min(meta["P_local"], (1 << log2_rays) // fw))
    cpu = {k: v.detach().cpu() for k, v in args.items()}
    cpu["x"], cpu["y"] = cpu["x"][:, :, :p].contiguous(), cpu["y"][:, :, :p].contiguous()
    names = tuple(k for k in LEAF_NAMES if k in cpu)
    leaves = [cpu[k].requires_grad_(True) for k in names]
    is_asph = "kappa" in cpu
    kind = ((cpu["kappa"].reshape(-1) != 0) | (cpu["poly"].reshape(-1, 4) != 0).any(dim=1)).int().tolist() if is_asph else None

    def one(dt=None, ieee=False):
        src = cpu if dt is None else {k: (v.to(dt) if v.is_floating_point() else v) for k, v in cpu.items()}
        lv = leaves if dt is None else [src[k].detach().requires_grad_(True) for k in names]
        if dt is not None:
            src.update(dict(zip(names, lv)))
        for q in lv:
            q.grad = None
        if is_asph:
            x, y, cx, cy, ok, back, _ = orc.trace_skew_general(
                src["x"], src["y"], src["z"], src["cx"], src["cy"], src["c"], src["t"], src["mu"], src["mask"],
                src["kappa"].reshape(-1), src["poly"].reshape(-1, 4), kind, ieee_sqrt=ieee)
        else:
            x, y, cx, cy, ok, back = orc.trace_skew(src["x"], src["y"], src["z"], src["cx"], src["cy"], src["c"],
                                                    src["t"], src["mu"], src["mask"], ieee_sqrt=ieee)
        orc.compute_rms2d(x, y, ok).backward()
        return [q.grad.clone() for q in lv]
    t0 = time.perf_counter()
    one()                                   # warm-up (first touch is ~15x slower)
    print(f"[bench] cpu_baseline warm-up {time.perf_counter() - t0:.1f}s", file=sys.stderr, flush=True)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        g32 = one()
        times.append(time.perf_counter() - t0)
        print(f"[bench] cpu_baseline rep {times[-1]:.2f}s", file=sys.stderr, flush=True)
    med = sorted(times)[1]
    g64 = one(torch.float64)
    g32i = one(None, ieee=True)
    # the same sample through the HIP path
    dev = args["x"].device
    gl = [args[k].detach().clone().requires_grad_(True) for k in names]
    ga = dict(args)
    ga.update(dict(zip(names, gl)))
    x, y, cx, cy, ok, back = ta.trace_skew(args["x"][:, :, :p].contiguous(), args["y"][:, :, :p].contiguous(), ga["z"],
                                           ga["cx"], ga["cy"], ga["c"], ga["t"], ga["mu"], ga["mask"], mode=mode,
                                           **{k: ga[k] for k in ("kappa", "poly") if k in ga})
    ta.compute_rms2d(x, y, ok).backward()

    def rel(a, b):
        return float(((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300)).item())
    gc = {}
    for k, q, r32, r32i, r64 in zip(names, gl, g32, g32i, g64):
        gc[k] = dict(vs_fp32_autograd_ieee_sqrt=rel(q.grad.cpu(), r32i), vs_fp32_autograd_mkl_sqrt=rel(q.grad.cpu(), r32),
                     vs_fp64_autograd=rel(q.grad.cpu(), r64), fp32_autograd_mkl_vs_ieee=rel(r32, r32i),
                     fp32_autograd_mkl_vs_fp64=rel(r32, r64))
    lens_groups = tuple(k for k in ("c", "t", "mu", "kappa", "poly") if k in gc)
    grad_check = dict(sample_rays=p * fw, arith_mode=mode, per_group=gc,
                      max_vs_fp32_autograd=max(gc[k]["vs_fp32_autograd_ieee_sqrt"] for k in lens_groups),
                      max_vs_fp32_autograd_mkl_sqrt=max(gc[k]["vs_fp32_autograd_mkl_sqrt"] for k in lens_groups),
                      max_vs_fp64_autograd=max(gc[k]["vs_fp64_autograd"] for k in lens_groups),
                      pytorch_fp32_self_noise=max(gc[k]["fp32_autograd_mkl_vs_ieee"] for k in lens_groups),
                      note="norm-relative error per parameter group, max over the lens parameters c, t, mu. "
                           "PyTorch fp32 autograd = the CPU oracle (bit-exact with the reference on CPU). torch.sqrt on "
                           "CPU (MKL) is up to 1 ulp off; 'ieee_sqrt' is the same autograd graph with a correctly "
                           "rounded sqrt. pytorch_fp32_self_noise = how far those two PyTorch runs are from each "
                           "other: the floor below which 'vs PyTorch autograd' is not defined for this lens")
    # SURVEY 8(d), optional extra line: the same eager graph (oracle, ~86 elementwise launches per surface +
    # autograd) on the MI355X itself, i.e. fused kernels vs eager PyTorch on identical silicon
    eager_gpu = None
    if not is_asph:
        try:
            pg = max(1, min(meta["P_local"], (1 << 22) // fw))      # 4 M rays: past the launch-bound regime,
            gsrc = {k: (v.detach().to(dev)) for k, v in cpu.items()}     # ~13 GB of autograd-saved tensors at 11 rows
            gsrc["x"], gsrc["y"] = args["x"][:, :, :pg].detach().contiguous(), args["y"][:, :, :pg].detach().contiguous()
            glv = [gsrc[k].requires_grad_(True) for k in names]

            def one_gpu():
                for q in glv:
                    q.grad = None
                o = orc.trace_skew(gsrc["x"], gsrc["y"], gsrc["z"], gsrc["cx"], gsrc["cy"], gsrc["c"], gsrc["t"],
                                   gsrc["mu"], gsrc["mask"])
                orc.compute_rms2d(o[0], o[1], o[4]).backward()
                torch.cuda.synchronize()
            one_gpu()
            tg = []
            for _ in range(3):
                t0 = time.perf_counter()
                one_gpu()
                tg.append(time.perf_counter() - t0)
            eager_gpu = dict(value=pg * fw / sorted(tg)[1] / 1e6, unit="M rays/s",
                             note=f"the oracle's eager PyTorch graph + autograd on this MI355X, {pg * fw} rays of the same workload")
            del gsrc, glv
            torch.cuda.empty_cache()
        except Exception as e:                      # informational only: never fail the bench on it
            eager_gpu = dict(value=None, note=f"not measured: {type(e).__name__}: {e}"[:200])
    try:
        model = [ln.split(":")[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except Exception:
        model = "unknown"
    base = dict(value=p * fw / med / 1e6, unit="M rays/s", cores=cores, kind="port", eager_pytorch_on_gpu=eager_gpu,
                sample=f"{p * fw} rays ({p} pupil points x {fw} field-wavelengths) of the same workload, fwd+bwd, "
                       f"median of 3 after 1 warm-up, torch threads={cores}, cpu='{model}'")
    return base, grad_check


if __name__ == "__main__":
    main()
