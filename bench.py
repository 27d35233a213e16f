#!/usr/bin/env python3
"""
bench.py -- rays/s of the sequential ray-trace hot path, forward + backward, on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg3a|cfg2|cfg5] [--mode strict|fast]

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
(one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

A "step" is one pass of the hot path over one resident batch of rays: trace_skew (fused forward
kernel, per-ray outputs materialised as the reference API returns them, spot moments fused) ->
compute_rms2d (closed form on the moments; one tiny all-reduce when N > 1) -> backward (walk-back
adjoint kernel) -> gradients of the loss w.r.t. the trace parameters (c, t, mu, z, cy) [-> one
tiny all-reduce when N > 1].  The pupil coordinates are resident in HBM before the timed region.
The K-step region is timed `--repeats` times (default 3) after ONE warm-up; `value` is the median.

Workloads:
  cfg3a (default, the headline): BASELINE configs[2] AS WRITTEN -- synthesized double Gauss, 11 rows (10 refracting
        surfaces + stop), 2 of them aspheric (conic + a4, a6; Newton intersection), F=1 field (0.707), W=1 ('d'),
        circular pupil grid 4096 x 4096 = 2^24 rays PER GPU (weak scaling; at N=8 this is cfg4's 2^27 rays).
        Aspheres are an extension beyond the reference (parity unpinned by it, pinned by the FD-checked oracle and by
        analytic cases: stigmatic conic, Fermat).
  cfg3: the ALL-SPHERICAL variant of the same lens -- the arithmetic the reference itself pins (bit-exact forward);
        reported with its own roofline and gradient check under also.cfg3 of the default run.
  cfg3s: cfg3a with STRONG aspheres (sag departure 0.32 / 0.11 mm, three Newton evaluations per row instead of two).
  cfg2: Cooke triplet (7 rows), 1024 x 1024 pupil, 3 fields, W=1.
  cfg5: 20-row synthetic zoom, 5 fields x 3 wavelengths, 1024 x 1024 pupil.
  sweep (default run, N=1): {2^20, 2^22, 2^24, 2^26} rays x {7, 11, 20} rows, F=W=1, fwd+bwd.
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

LEAF_NAMES = ("z", "cy", "c", "t", "mu", "kappa", "poly")   # differentiable arguments of trace_skew
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: peak FP32 vector (every issue slot an FMA)
PMC_FILES = ("r03_pmc_traffic.json", "r02_pmc_traffic.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=3, help="how often the K-step region is timed (value = median)")
    ap.add_argument("--workload", default="cfg3a", choices=["cfg3a", "cfg3", "cfg3s", "cfg2", "cfg5"])
    ap.add_argument("--mode", default=os.environ.get("TORCHOPTICS_AMD_MODE", "strict"), choices=["strict", "fast"])
    ap.add_argument("--log2-pupil", type=int, default=None, help="override log2 of pupil points per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse several ranks on one GPU)")
    ap.add_argument("--force-dist", action="store_true",
                    help="with one process: still build a 1-rank process group, so both collectives of the sharded "
                         "path execute (RCCL on one GPU)")
    ap.add_argument("--no-other-mode", action="store_true", help="skip the secondary measurement of the other arithmetic mode")
    ap.add_argument("--no-also", action="store_true", help="skip the extra workloads reported next to the main one")
    ap.add_argument("--no-sweep", action="store_true", help="skip the rays x rows sweep")
    ap.add_argument("--no-fp64-check", action="store_true", help="skip the full-size comparison with the double-precision kernels")
    ap.add_argument("--graph", action="store_true",
                    help="record one step (both kernels + reductions + closed form + autograd bookkeeping) into a HIP "
                         "graph and time replays of it instead of eager steps")
    ap.add_argument("--no-graph-child", action="store_true", help="skip the secondary HIP-graph measurement (child process)")
    ap.add_argument("--cpu-log2-rays", type=int, default=20, help="log2 of the gradient-check sample (rays)")
    ap.add_argument("--cpu-full-log2-rays", type=int, default=24,
                    help="log2 of the rays of the timed CPU baseline (chunked at 2^22 rays)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------ workloads
def build_lens(name, device):
    from torchoptics_amd import prescriptions as P
    if name in ("cfg3", "cfg3a", "cfg3s", "dg11"):
        return P.double_gauss(device, aspheres={"cfg3a": True, "cfg3s": "strong"}.get(name, False))
    if name in ("cfg2", "cooke7"):
        import yaml_free_lenses as L
        return L.build("cooke", device)
    return P.zoom20(device)


def workload(name, device, world, rank, log2_pupil, fields=None, wl=None):
    """Returns (trace_skew arguments as leaves, meta, extras) for this rank's pupil slice."""
    import torchoptics_amd as ta
    from torchoptics_amd import ray_tracing as rt
    lens, specs, leaves = build_lens(name, device)
    dflt = {"cfg3": ((0.707,), ("d",), 24), "cfg3a": ((0.707,), ("d",), 24), "cfg3s": ((0.707,), ("d",), 24),
            "cfg2": ((0., 0.707, 1.), ("d",), 20),
            "cfg5": (tuple(np.linspace(0, 1, 5)), ("C", "d", "F"), 20)}.get(name, ((0.707,), ("d",), 24))
    fields = fields or dflt[0]
    wl = wl or dflt[1]
    lp = log2_pupil if log2_pupil is not None else dflt[2]
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
    n_asph = 0
    if "kappa" in args:
        n_asph = int(((args["kappa"].reshape(-1) != 0) | (args["poly"].reshape(-1, 4) != 0).any(dim=1)).sum().item())
    meta = dict(F=len(fields), W=len(wl), S=lens.c.shape[1], P_local=p_local, P_total=p_local * world,
                lens=name, fields=list(map(float, fields)), wavelengths=list(wl), n_asph=n_asph,
                n_r=n_r, n_theta=n_theta_total)
    return args, meta, (tr, specs, lens, leaves, xy)


class Job:
    """One workload resident on this rank's GPU and the step that is timed."""

    def __init__(self, name, device, world, rank, group, log2_pupil=None, fields=None, wl=None, penalty_rate=None):
        self.args, self.meta, self.extra = workload(name, device, world, rank, log2_pupil, fields, wl)
        self.name, self.group, self.world = name, group, world
        self.penalty_rate = penalty_rate        # not None: the real caller's loss rms + penalty_rate * sumQ (aggregate='sum')
        self.leaves = [self.args[k] for k in LEAF_NAMES if k in self.args]
        self.asph = {k: self.args[k] for k in ("kappa", "poly") if k in self.args}
        self.n_per_field_total = self.meta["P_total"] * self.meta["W"]
        self.rays_local = self.meta["F"] * self.meta["W"] * self.meta["P_local"]
        self.rays_total = self.rays_local * world

    def step(self):
        import torchoptics_amd as ta
        from torchoptics_amd import dist as tl_dist
        a = self.args
        for p in self.leaves:
            p.grad = None
        if self.penalty_rate is not None:
            # optics_simulator_lite.py:430-450: loss_unsup = rms + penalty_rate * sumQ, sumQ from the fused penalty sums
            out = ta.trace_skew(a["x"], a["y"], a["z"], a["cx"], a["cy"], a["c"], a["t"], a["mu"], a["mask"], "sum", **self.asph)
            rms = ta.compute_rms2d(out[0], out[1], out[4], group=self.group, n_per_field=self.n_per_field_total)
            (rms + self.penalty_rate * ta.ray_tracing.penalty_sum(out[6], self.meta["S"])).backward()
            return rms.detach()
        x, y, cx, cy, ok, back = ta.trace_skew(a["x"], a["y"], a["z"], a["cx"], a["cy"], a["c"], a["t"], a["mu"],
                                               a["mask"], **self.asph)
        rms = ta.compute_rms2d(x, y, ok, group=self.group, n_per_field=self.n_per_field_total)
        rms.backward()
        if self.group is not None:
            tl_dist.all_reduce_grads(self.leaves, self.group)
        return rms.detach()


_T0 = time.perf_counter()


def log(msg):
    """Progress line on stderr (the JSON line on stdout stays the only stdout output)."""
    print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def sync(group):
    if group is not None:
        torch.distributed.barrier(group)
    torch.cuda.synchronize()


def timed(job, mode, steps, warmup, repeats, graph, backend, device):
    """W untimed warm-up steps, then `repeats` regions of exactly K steps, each between barrier+synchronize;
    every region's time is the max over ranks.  Returns (list of seconds, kernel ms dict, rms)."""
    from torchoptics_amd import ops
    ops.set_default_mode(mode)
    group = job.group
    times, km = [], {}
    if graph:
        # warm up ON the capture stream: every autograd node and allocation of the step is created there
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            for _ in range(max(warmup, 3)):
                r = job.step()
        torch.cuda.current_stream().wait_stream(cap)
        for p_ in job.leaves:
            p_.grad = None
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            r = job.step()
        for _ in range(warmup):
            g.replay()
        for _ in range(repeats):
            sync(group)
            t0 = time.perf_counter()
            for _ in range(steps):
                g.replay()
            sync(group)
            times.append(time.perf_counter() - t0)
    else:
        for _ in range(warmup):
            r = job.step()
        for _ in range(repeats):
            sync(group)
            ops.enable_timing(True)
            t0 = time.perf_counter()
            for _ in range(steps):
                r = job.step()
            sync(group)
            times.append(time.perf_counter() - t0)
            k = ops.timing_ms()
            ops.enable_timing(False)
            for key, v in k.items():
                km.setdefault(key, []).append(v)
        km = {key: (statistics.median(v) if all(q is not None for q in v) else None) for key, v in km.items()}
    if group is not None:
        tmax = torch.tensor(times, dtype=torch.float64, device="cpu" if backend == "gloo" else device)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX, group=group)
        times = tmax.tolist()
    return times, km, r


# ------------------------------------------------------------------------------------------ roofline
def pmc_traffic(workload_name, mode, kernel, meta):
    """HBM bytes per launch measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes,
    gfx950 FETCH correction applied) for this workload at its default size; None if not profiled."""
    if meta["P_local"] != (1 << 24):
        return None, None
    for fn in PMC_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", fn)) as f:
                d = json.load(f)
            return d[workload_name][mode][kernel]["hbm_bytes"], fn
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def flops_per_ray(S, n_asph=0):
    """fp32 arithmetic operations per ray as written in csrc/tl_kernels.inc (mul, add/sub, sqrt, div,
    rcp, rsq each count 1, an FMA-able pair counts 2; compares, selects, negations, abs not counted; counted by
    hand from the source, see DESIGN.md "Flop counts").
      spherical row : step_fwd 54 | walk-back = inverse refraction 20 + intersection quantities shared between the
                      forward step's e, m2, tmp and the back-intersection with the previous surface 47 + 3 reciprocals
                      + adjoint with the per-row sums 82 = 152 | checkpoint backward = forward sweep 54 + recompute 49
                      + adjoint 84
      aspheric row  : step_fwd_asph = sphere guess 27 + 2 Newton steps x 53 (the minimum: one step, plus the one
                      after the convergence vote) + converged sag / acceptance test 43 + vector Snell 36 = 212;
                      walk-back = aspheric normal + inverse vector Snell 55, Newton hit on that row from the next one
                      27 + 2 x 53 + 6 = 139, step_bwd_asph 238 + coefficient wave sums 14 = 446;
                      checkpoint backward = forward sweep 212 + step_bwd_asph 252
    The Newton step count is data dependent; 2 per row is the floor, so aspheric flop rates are LOWER bounds.
    (Round 2, second half: the adjoint was re-written with its factors of 2 and 1/2 folded by hand, -15 operations per
    row, and the walk-back shares one dot product and one |h|^2 between the two intersections, -3: 167 -> 152 and
    202 -> 187 per row; rates quoted before that change used the old counts.)"""
    sph = S - n_asph
    fwd = 54 * sph + 212 * n_asph + 5
    bwd_ck = fwd + (49 + 84) * sph + 252 * n_asph + 30
    # round 3: the walk-back over an aspheric row reads the forward's stored hit instead of iterating (139 -> sag at the
    # stored point 17 + distance along the line 8 = 25) and feeds the adjoint with the normal it already has
    # (step_bwd_asph 238 -> asph_adjoint 186; aspheric normal + inverse vector Snell 47; quad sums of the five
    # coefficient terms 8): 266 per aspheric row, whatever the strength of the asphere
    bwd_inv = 152 * sph + 266 * n_asph + 40
    return fwd, bwd_ck, bwd_inv


def roofline_of(job, kern_ms, mode):
    """roofline / roofline_valu / per-kernel table of one measured workload (event-timed launches)."""
    from torchoptics_amd import ops
    meta = job.meta
    fw = meta["F"] * meta["W"]
    inv = ops.get_backward_algorithm() == "inverse"
    # algorithmic bytes per ray (DESIGN.md "bytes per unit"): forward writes x,y,cx,cy,ok,back and reads x_in,y_in;
    # the walk-back backward reads x_in,y_in and the forward's x,y,cx,cy,ok; the checkpoint backward only x_in,y_in
    # (+ 8 bytes per ray and aspheric row each way: the hit points the forward leaves for the walk-back, tl_problem.asph_hits)
    hit_b = 8.0 * min(meta["n_asph"], ops.ASPH_HIT_SLOTS) if (inv and meta["n_asph"] <= ops.ASPH_HIT_SLOTS) else 0.0
    b_fwd, b_bwd = 18.0 + 8.0 / fw + hit_b, (17.0 + hit_b if inv else 0.0) + 8.0 / fw
    f_fwd, f_ck, f_inv = flops_per_ray(meta["S"], meta["n_asph"])
    f_bwd = f_inv if inv else f_ck
    # the walk-back kernel launch_bwd_inv picks (csrc/tl_kernels.inc: kInvUnrollMin / kInvUnrollMax)
    unrolled = 3 <= meta["S"] <= 20 and meta["P_local"] >= 256 and os.environ.get("TL_INV_ROLLED") != "1"
    if not inv:
        bwd_name = "trace_bwd_kernel"
    elif unrolled and (not meta["n_asph"] or hit_b):
        bwd_name = f"trace_bwd_inv_unrolled_kernel<{meta['S']}, {'true' if meta['n_asph'] else 'false'}, false>"
    else:
        bwd_name = f"trace_bwd_inv_kernel<{'true' if meta['n_asph'] else 'false'}>"
    kernels = {}
    for key, bpr, fpr in (("fwd", b_fwd, f_fwd), ("bwd", b_bwd, f_bwd)):
        ms = kern_ms.get(key)
        if ms:
            kernels[key] = dict(ms=ms, rays_per_s=job.rays_local / ms * 1e3, hbm_GBs=job.rays_local * bpr / ms / 1e6,
                                valu_TFLOPs=job.rays_local * fpr / ms / 1e9, bytes_per_ray=bpr, flops_per_ray=fpr)
    dom = kernels.get("bwd")
    if not dom:
        return None, None, kernels, b_fwd + b_bwd
    traffic, src = pmc_traffic(job.name, mode, "bwd", meta)
    roofline = dict(kernel=bwd_name, bound="hbm", achieved=dom["hbm_GBs"], peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=dom["hbm_GBs"] / HBM_PEAK_GBS, traffic=traffic, traffic_source=src,
                    launch_ms=dom["ms"], algorithmic_bytes_per_launch=job.rays_local * b_bwd,
                    note="per-ray FMA kernel: the binding limit is the FP32 vector ALU, see roofline_valu; traffic = "
                         "HBM bytes per launch from the committed rocprofv3 PMC passes, null for sizes not profiled")
    valu = dict(kernel=bwd_name, bound="valu_fp32", achieved=dom["valu_TFLOPs"], peak=VALU_PEAK_TFLOPS, unit="TFLOP/s",
                frac=dom["valu_TFLOPs"] / VALU_PEAK_TFLOPS,
                note="walk-back over stored hits: no Newton iteration in this kernel, the count is exact" if meta["n_asph"] else None)
    return roofline, valu, kernels, b_fwd + b_bwd


def grad_report(job, group, backend, device):
    """The leaf gradients the last step left behind (after the gradient all-reduce when sharded): values as seen by
    rank 0, and whether every rank holds the same bits (all-gather of the packed vector, compared on every rank)."""
    names = [k for k in LEAF_NAMES if k in job.args and job.args[k].grad is not None]
    flat = torch.cat([job.args[k].grad.detach().reshape(-1).double() for k in names])
    same = True
    if group is not None:
        world = torch.distributed.get_world_size(group)
        src = flat.cpu() if backend == "gloo" else flat
        got = [torch.empty_like(src) for _ in range(world)]
        torch.distributed.all_gather(got, src, group=group)
        same = all(torch.equal(g, got[0]) for g in got)
    return dict(bitwise_equal_across_ranks=bool(same),
                values={k: job.args[k].grad.detach().reshape(-1).cpu().tolist() for k in names})


def full_size_fp64_check(job):
    """The gradients of the last fp32 step against the SAME fan traced by the double-precision kernels
    (tl_trace_fwd_f64 / tl_trace_bwd_f64: generic fp64 twins of the trace, checked against the oracle's fp64 autograd in
    tests/test_gpu_f64.py) at the FULL size of the workload -- the CPU oracle's gradient check stops at 2^20 rays."""
    import torchoptics_amd as ta
    names = [k for k in LEAF_NAMES if k in job.args and job.args[k].grad is not None]
    a64 = {k: (v.detach().double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in job.args.items()}
    for k in names:
        a64[k].requires_grad_(True)
    extra = {k: a64[k] for k in ("kappa", "poly") if k in a64}
    t0 = time.perf_counter()
    x, y, cx, cy, ok, back = ta.trace_skew(a64["x"], a64["y"], a64["z"], a64["cx"], a64["cy"], a64["c"], a64["t"], a64["mu"],
                                           a64["mask"], **extra)
    rms = ta.compute_rms2d(x, y, ok, group=job.group, n_per_field=job.n_per_field_total)
    rms.backward()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    def rel(p, q):
        return float(((p.double() - q).norm() / q.norm().clamp_min(1e-300)).item())
    per = {k: rel(job.args[k].grad, a64[k].grad) for k in names}
    lens_groups = [k for k in ("c", "t", "mu", "kappa", "poly") if k in per]
    return dict(rays=job.rays_local, rms_fp64=float(rms.item()), per_group=per, max_lens_parameters=max(per[k] for k in lens_groups),
                fp64_step_ms=dt * 1e3,
                note="norm-relative distance of the fp32 step's gradients from the double-precision kernels' on the whole fan "
                     "of this rank (before the gradient all-reduce when sharded); z, cy are residual-type gradients")


def summarize(job, times, steps):
    med = statistics.median(times)
    return dict(value=job.rays_total * steps / med / 1e6, unit="M rays/s", ms_per_step=med / steps * 1e3,
                repeats_ms_per_step=[t / steps * 1e3 for t in times])


# ------------------------------------------------------------------------------------------ main
def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks ourselves (torch.distributed.run, one process per
        # GPU) as a child process tree BEFORE this process has made any GPU call, pass rank 0's JSON line through
        # and leave with the child's exit code
        n_dev = torch.cuda.device_count()              # (counting devices does not initialise HIP)
        if a.backend == "nccl" and a.gpus > n_dev:
            raise SystemExit(f"--gpus {a.gpus} but this node shows {n_dev} GPU(s): RCCL needs one GPU per rank "
                             "(--backend gloo rehearses several ranks on one GPU)")
        from torchoptics_amd import dist as tl_dist
        sys.stdout.flush()
        raise SystemExit(tl_dist.spawn_local_ranks(os.path.abspath(__file__), sys.argv[1:], a.gpus))
    assert torch.cuda.is_available(), "bench.py needs an AMD GPU"
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = f"cuda:{local}"

    from torchoptics_amd import _lib, ops, dist as tl_dist
    group = tl_dist.init_group(device, a.backend, force=a.force_dist)
    n_seen = tl_dist.ranks_seen(group, "cpu" if a.backend == "gloo" else device) if group is not None else 1
    _lib.lib()      # fail loudly when the HIP library is missing
    ops.set_default_mode(a.mode)

    log(f"workload {a.workload}, mode {a.mode}, world {world}, graph {a.graph}")
    job = Job(a.workload, device, world, rank, group, a.log2_pupil)
    meta = job.meta
    times, kern_ms, rms = timed(job, a.mode, a.steps, a.warmup, a.repeats, a.graph, a.backend, device)
    head = summarize(job, times, a.steps)
    final_grads = grad_report(job, group, a.backend, device)
    fp64_full = None
    if world == 1 and not a.graph and not a.no_fp64_check:
        try:
            job.step()                       # fresh fp32 gradients of one step
            fp64_full = full_size_fp64_check(job)
        except Exception as e:               # a checker problem must not cost the measurement
            fp64_full = dict(error=f"{type(e).__name__}: {e}"[:300])
        torch.cuda.empty_cache()
    roofline, roofline_valu, kernels, step_bpr = roofline_of(job, kern_ms, a.mode)
    inv = ops.get_backward_algorithm() == "inverse"
    solo = world == 1

    log(f"main: {head['value']:.0f} M rays/s")
    other = None
    if not a.no_other_mode:
        om = "fast" if a.mode == "strict" else "strict"
        ot, okm, orms = timed(job, om, a.steps, a.warmup, a.repeats, a.graph, a.backend, device)
        other = dict(arith_mode=om, **summarize(job, ot, a.steps), fwd_kernel_ms=okm.get("fwd"),
                     bwd_kernel_ms=okm.get("bwd"), rms=float(orms.item()))
        ops.set_default_mode(a.mode)

    # secondary workloads, same protocol, reported under "also" (N = 1 only: they are not part of the scaling run)
    also = {}
    if solo and not a.no_also and a.workload == "cfg3a" and a.log2_pupil is None:
        for wname in ("cfg3", "cfg3s", "cfg2", "cfg5"):
            log(f"also: {wname}")
            j2 = Job(wname, device, world, rank, group)
            t2, km2, r2 = timed(j2, a.mode, a.steps, a.warmup, a.repeats, a.graph, a.backend, device)
            e = dict(**summarize(j2, t2, a.steps), fwd_kernel_ms=km2.get("fwd"), bwd_kernel_ms=km2.get("bwd"),
                     rays=j2.rays_total, rows=j2.meta["S"], F=j2.meta["F"], W=j2.meta["W"], rms=float(r2.item()),
                     arith_mode=a.mode, hip_graph=bool(a.graph))
            if km2.get("fwd") and km2.get("bwd"):
                # small workloads are bound by the eager Python/autograd chain, not by the GPU: what the two trace
                # kernels alone sustain, and (hip_graph_value, from the graph child) what a recorded step reaches
                e["trace_kernels_only_value"] = j2.rays_total / (km2["fwd"] + km2["bwd"]) / 1e3
            if wname in ("cfg3", "cfg3s") and not a.graph:
                # the all-spherical variant (the arithmetic the reference pins) and the strong-asphere variant: the
                # full evidence of a headline line
                rf, rv, kk, _ = roofline_of(j2, km2, a.mode)
                e.update(roofline=rf, roofline_valu=rv, kernels=kk, workload=workload_label(wname, j2.meta, j2))
                if not a.no_cpu_baseline:
                    log(f"also: {wname} gradient check against the oracle (CPU)")
                    _, e["grad_rel_err_vs_pytorch_autograd"] = cpu_leg(j2.args, j2.meta, min(a.cpu_log2_rays, 18), a.mode,
                                                                       time_it=False)
            also[wname] = e
            del j2
            torch.cuda.empty_cache()

    # the real caller's loss (rms + 0.2 sumQ, aggregate='sum') on the headline lens and on its all-spherical variant
    if solo and not a.no_also and a.workload == "cfg3a" and a.log2_pupil is None:
        for wname in ("cfg3a", "cfg3"):
            log(f"also: {wname} with the penalty term")
            j2 = Job(wname, device, world, rank, group, penalty_rate=0.2)
            t2, km2, r2 = timed(j2, a.mode, a.steps, a.warmup, a.repeats, a.graph, a.backend, device)
            also[wname + "_full_loss"] = dict(**summarize(j2, t2, a.steps), fwd_kernel_ms=km2.get("fwd"), bwd_kernel_ms=km2.get("bwd"),
                                              rays=j2.rays_total, rows=j2.meta["S"], rms=float(r2.item()), arith_mode=a.mode,
                                              hip_graph=bool(a.graph),
                                              workload=f"{wname}: loss_unsup = rms + 0.2 sumQ (optics_simulator_lite.py:430-450), "
                                                       "aggregate='sum', fwd+bwd; backward = walk-back of the live rays + checkpoint "
                                                       "pass over the rays that died on the way")
            del j2
            torch.cuda.empty_cache()

    # north_star sweep: 1-64 M rays x 7 / 11 / 20 rows, one field, one wavelength, fwd+bwd
    sweep = None
    if solo and not a.no_sweep and a.workload == "cfg3a" and a.log2_pupil is None:
        sweep = []
        for lens_name, rows in (("cooke7", 7), ("dg11", 11), ("zoom20", 20)):
            log(f"sweep: {rows} rows")
            for lp in (20, 22, 24, 26):
                j3 = Job(lens_name, device, 1, 0, None, lp, fields=(0.707,), wl=("d",))
                t3, km3, r3 = timed(j3, a.mode, max(5, a.steps // 2), 2, 1, a.graph, a.backend, device)
                s3 = summarize(j3, t3, max(5, a.steps // 2))
                sweep.append(dict(rows=rows, rays=j3.rays_total, value=s3["value"], ms_per_step=s3["ms_per_step"],
                                  fwd_kernel_ms=km3.get("fwd"), bwd_kernel_ms=km3.get("bwd"),
                                  hbm_GBs=j3.rays_total * step_bpr_of(j3) / (s3["ms_per_step"] * 1e6)))
                del j3
                torch.cuda.empty_cache()

    # the same step replayed from a HIP graph, measured in a CHILD process (a capture failure of the
    # PyTorch/ROCm stack must not cost the main line); N = 1 only
    hip_graph = None
    if solo and group is None and not a.graph and not a.no_graph_child:
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--graph", "--steps", str(a.steps), "--warmup", str(a.warmup),
               "--repeats", str(a.repeats), "--workload", a.workload, "--mode", a.mode, "--no-cpu-baseline",
               "--no-other-mode"]
        if a.no_also:
            cmd += ["--no-also"]
        if a.no_sweep:
            cmd += ["--no-sweep"]
        if a.log2_pupil is not None:
            cmd += ["--log2-pupil", str(a.log2_pupil)]
        log("HIP-graph child process")
        try:
            cp = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            line = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
            if cp.returncode == 0 and line:
                child = json.loads(line[-1])
                hip_graph = dict(value=child["value"], unit="M rays/s", ms_per_step=child["ms_per_step"],
                                 repeats_ms_per_step=child["repeats_ms_per_step"],
                                 note="the identical step recorded once into a HIP graph and replayed")
                for wname, e in (child.get("also") or {}).items():
                    if wname in also:
                        also[wname]["hip_graph_value"] = e["value"]
                        also[wname]["hip_graph_ms_per_step"] = e["ms_per_step"]
                for e, ce in zip(sweep or [], child.get("sweep") or []):
                    if (e["rows"], e["rays"]) == (ce["rows"], ce["rays"]):
                        e["hip_graph_value"] = ce["value"]
                        e["hip_graph_ms_per_step"] = ce["ms_per_step"]
            else:
                hip_graph = dict(error=f"child exited with {cp.returncode}", stderr_tail=cp.stderr[-300:])
        except Exception as e:       # timeout or launch failure
            hip_graph = dict(error=repr(e))

    # BASELINE configs[4]: the 100-step Adam loop on the 20-row zoom (5 fields x 3 wavelengths, 2^20 pupil points per
    # GPU), every step = Lens assembly + dispersion + pupil position + forward + RMS + backward + Adam; eager and with
    # the whole step replayed from a HIP graph (child processes: examples/adam_loop.py is the harness)
    if solo and group is None and not a.graph and not a.no_also and a.workload == "cfg3a" and a.log2_pupil is None:
        import subprocess
        log("cfg5 Adam loop (child processes)")
        adam = {}
        for tag, flags in (("eager", []), ("hip_graph", ["--graph"])):
            try:
                cp = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "adam_loop.py"), "--steps", "100",
                                     "--log2-pupil", "20", "--mode", a.mode] + flags, capture_output=True, text=True, timeout=240)
                line = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
                r = json.loads(line[-1]) if cp.returncode == 0 and line else None
                adam[tag] = (dict(steps_per_s=r["steps_per_s"], M_rays_per_s=r["M_rays_per_s"], loss_initial=r["loss_initial"],
                                  loss_final=r["loss_final"]) if r else dict(error=f"exit {cp.returncode}", stderr_tail=cp.stderr[-300:]))
            except Exception as e:
                adam[tag] = dict(error=repr(e))
        also["cfg5_adam_loop"] = dict(workload="20-row zoom, F=5 W=3 P=2^20 (15.7 M rays per step), 100 Adam steps on c and t, "
                                               "1 GPU; steps include the whole host chain", arith_mode=a.mode, **adam)

        # the reference's real caller as a minibatch: 256 lenses x 1 536 rays, aggregate + ray aiming, per-lens losses;
        # one batched launch each way against the caller's one-lens-at-a-time loop (examples/minibatch_loss.py)
        log("lens minibatch (child process)")
        try:
            cp = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "minibatch_loss.py"), "--lenses", "256", "--steps", "20",
                                 "--loop-lenses", "16", "--graph", "--mode", a.mode], capture_output=True, text=True, timeout=240)
            line = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
            also["lens_minibatch"] = (json.loads(line[-1]) if cp.returncode == 0 and line
                                      else dict(error=f"exit {cp.returncode}", stderr_tail=cp.stderr[-300:]))
        except Exception as e:
            also["lens_minibatch"] = dict(error=repr(e))

    cpu_baseline, grad_check, leaf_grads = None, None, None
    if rank == 0 and solo and not a.no_cpu_baseline:
        log("cpu baseline + gradient check")
        cpu_baseline, grad_check = cpu_leg(job.args, meta, a.cpu_log2_rays, a.mode, full_log2=a.cpu_full_log2_rays)
        log("leaf gradient check")
        leaf_grads = leaf_grad_check(a.workload, device, a.mode)
        log("done")

    if rank == 0:
        out = {
            "metric": "M rays/s through 10-surface lens; fwd+bwd; grad rel-err vs PyTorch autograd",
            "value": head["value"], "unit": "M rays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "repeats": a.repeats, "repeats_ms_per_step": head["repeats_ms_per_step"], "n_ranks_seen": n_seen,
            "config": {"workload": workload_label(a.workload, meta, job),
                       "arith_mode": a.mode, "backward_algorithm": "walk-back from the forward outputs (checkpoint kernel "
                       "as on-device fallback for ill-conditioned fans)" if inv else "checkpoint",
                       "parallelism": f"pupil-sharded dp{world}",
                       "collectives": "none" if group is None else ("rccl" if a.backend == "nccl" else a.backend),
                       "collective_shape": None if group is None else tl_dist.get_collective(),
                       "rms": float(rms.item()), "hip_graph": bool(a.graph), "host_chain": ops.host_chain()},
            "roofline": roofline, "roofline_valu": roofline_valu, "kernels": kernels,
            "step_hbm_GBs": job.rays_total * step_bpr / (head["ms_per_step"] * 1e6),
            "grad_rel_err_vs_pytorch_autograd": grad_check,
            "leaf_grads": leaf_grads,
            "final_grads": final_grads,
            "grad_vs_fp64_kernels_full_size": fp64_full,
            "other_mode": other,
            "hip_graph_replay": hip_graph,
            "also": also,
            "sweep": sweep,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out))
    if group is not None:
        torch.distributed.destroy_process_group()


def step_bpr_of(job):
    fw = job.meta["F"] * job.meta["W"]
    return (18.0 + 8.0 / fw) + (17.0 + 8.0 / fw)


def workload_label(name, meta, job):
    what = {"cfg3": "ALL-SPHERICAL variant of BASELINE configs[2] (double Gauss, no aspheric rows: the arithmetic the "
                    "reference itself pins)",
            "cfg3a": "BASELINE configs[2] as written: double Gauss with 2 aspheric rows (aspheres are an extension: parity "
                     "unpinned by the reference, pinned by the oracle and analytic cases; the all-spherical variant is under also.cfg3)",
            "cfg3s": "BASELINE configs[2] with STRONG aspheres (sag departure 0.32 / 0.11 mm: 3 Newton evaluations per row)",
            "cfg2": "BASELINE configs[1]: Cooke triplet", "cfg5": "BASELINE configs[4] lens: 20-row zoom"}[name]
    return (f"{name}: {what}; S={meta['S']} rows, F={meta['F']} W={meta['W']} P={meta['P_local']} pupil points per GPU "
            f"({job.rays_local} rays/GPU, {job.rays_total} total), circular grid, loss=compute_rms2d, fwd+bwd")


# ------------------------------------------------------------------------------------------ CPU legs (the checker)
def _cpu_threads():
    """(threads to use, description).  Every core this process may really run on: the scheduler affinity, cut to
    the cgroup CPU quota when there is one -- a GPU box hands each job a share of its host cores, and an OpenMP
    team larger than that share spends its time spinning at barriers (observed: a 1 s oracle call not finishing
    in 7 minutes).  Without quota information the team is capped at 32 threads; TL_CPU_THREADS overrides."""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(float(q) / float(per) + 0.5))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, int(q / per + 0.5))
        except (OSError, ValueError):
            pass
    if os.environ.get("TL_CPU_THREADS"):
        n = max(1, min(aff, int(os.environ["TL_CPU_THREADS"])))
        why = "TL_CPU_THREADS"
    elif quota is not None:
        n, why = min(aff, quota), f"cgroup cpu quota {quota}"
    else:
        n, why = min(aff, 32), "no cgroup quota visible: capped at 32"
    return n, f"affinity {aff} cpus, {why}"


def cpu_leg(args, meta, log2_rays, mode, time_it=True, full_log2=24):
    """The oracle (CPU restatement of the reference, eager PyTorch fp32 + autograd) on this box's host cores.
    kind='port': the reference itself cannot travel to the GPU box.
      timing  (BASELINE.md section 3): every core the process may use, the first 2^full_log2 rays of the same
              workload in chunks of <= 2^22 rays (autograd keeps ~300 B/ray/surface), fwd+bwd per chunk,
              1 warm-up chunk + median of 3 passes;
      gradient check: the first 2^log2_rays rays traced by the HIP path and by the oracle in fp32 (MKL sqrt and
              correctly rounded sqrt) and fp64; norm-relative error per argument of trace_skew."""
    import torchoptics_amd as ta
    from oracle import trace_oracle as orc
    cores, cores_why = _cpu_threads()
    torch.set_num_threads(cores)
    fw = meta["F"] * meta["W"]
    p = max(1, min(meta["P_local"], (1 << log2_rays) // fw))
    cpu = {k: v.detach().cpu() for k, v in args.items()}
    x_all, y_all = cpu["x"], cpu["y"]
    names = tuple(k for k in LEAF_NAMES if k in cpu)
    leaves = [cpu[k].requires_grad_(True) for k in names]
    is_asph = "kappa" in cpu
    kind = ((cpu["kappa"].reshape(-1) != 0) | (cpu["poly"].reshape(-1, 4) != 0).any(dim=1)).int().tolist() if is_asph else None

    def one(lo, hi, dt=None, ieee=False):
        src = dict(cpu) if dt is None else {k: (v.to(dt) if v.is_floating_point() else v) for k, v in cpu.items()}
        src["x"], src["y"] = x_all[:, :, lo:hi].to(dt or x_all.dtype), y_all[:, :, lo:hi].to(dt or y_all.dtype)
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

    base = None
    if time_it:
        log(f"cpu_baseline: oracle on {cores} threads ({cores_why})")
        p_full = max(1, min(meta["P_local"], (1 << full_log2) // fw))
        chunk = max(1, (1 << 22) // fw)
        bounds = [(lo, min(lo + chunk, p_full)) for lo in range(0, p_full, chunk)]
        t0 = time.perf_counter()
        one(*bounds[0])                         # warm-up (first touch is ~15x slower)
        log(f"cpu_baseline warm-up chunk {time.perf_counter() - t0:.1f}s")
        passes = []
        for _ in range(3):
            t0 = time.perf_counter()
            for lo, hi in bounds:
                one(lo, hi)
            passes.append(time.perf_counter() - t0)
            log(f"cpu_baseline pass {passes[-1]:.2f}s")
            if sum(passes) > 45.0:              # bounded: never let the baseline dominate the run
                break
        med = statistics.median(passes)
        try:
            model = [ln.split(":")[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
        except Exception:
            model = "unknown"
        base = dict(value=p_full * fw / med / 1e6, unit="M rays/s", cores=cores, kind="port",
                    sample=f"{p_full * fw} rays of the same workload in {len(bounds)} chunks of <= {chunk * fw} rays "
                           f"(RMS loss and autograd backward per chunk), fwd+bwd, median of {len(passes)} passes after "
                           f"a 1-chunk warm-up, torch threads={cores} ({cores_why}), cpu='{model}'")

    log(f"gradient check: oracle fp32 / fp64 / fp32-ieee on {p * fw} rays, {cores} threads")
    g32 = one(0, p)
    g64 = one(0, p, torch.float64)
    g32i = one(0, p, None, ieee=True)
    log("gradient check: HIP path")
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
    launch = tuple(k for k in ("z", "cy") if k in gc)
    mx = lambda key, ks: max(gc[k][key] for k in ks)      # noqa: E731
    grad_check = dict(sample_rays=p * fw, arith_mode=mode, per_group=gc,
                      max_vs_fp32_autograd=mx("vs_fp32_autograd_ieee_sqrt", lens_groups),
                      max_vs_fp32_autograd_mkl_sqrt=mx("vs_fp32_autograd_mkl_sqrt", lens_groups),
                      max_vs_fp64_autograd=mx("vs_fp64_autograd", lens_groups),
                      pytorch_fp32_self_noise=mx("fp32_autograd_mkl_vs_ieee", lens_groups),
                      launch_conditions=dict(
                          max_vs_fp32_autograd=mx("vs_fp32_autograd_ieee_sqrt", launch),
                          max_vs_fp64_autograd=mx("vs_fp64_autograd", launch),
                          pytorch_fp32_self_noise=mx("fp32_autograd_mkl_vs_ieee", launch),
                          pytorch_fp32_vs_fp64=mx("fp32_autograd_mkl_vs_fp64", launch),
                          note="d/dz and d/dcy are residuals ~1e-3 of their per-ray terms: PyTorch's own two fp32 runs "
                               "differ by pytorch_fp32_self_noise here") if launch else None,
                      note="norm-relative error per argument group of trace_skew; max_* over the lens parameters "
                           "(c, t, mu[, kappa, poly]), the launch conditions z, cy are listed separately. PyTorch fp32 "
                           "autograd = the CPU oracle (bit-exact with the reference on CPU). torch.sqrt on CPU (MKL) is up "
                           "to 1 ulp off; 'ieee_sqrt' is the same autograd graph with a correctly rounded sqrt. "
                           "pytorch_fp32_self_noise = how far those two PyTorch runs are from each other: the floor "
                           "below which 'vs PyTorch autograd' is not defined for this lens")
    if base is not None and not is_asph:
        base["eager_pytorch_on_gpu"] = eager_on_gpu(orc, cpu, args, names, meta, dev)
    return base, grad_check


def eager_on_gpu(orc, cpu, args, names, meta, dev):
    """SURVEY 8(d), optional extra line: the same eager graph (oracle, ~86 elementwise launches per surface +
    autograd) on the MI355X itself, i.e. fused kernels vs eager PyTorch on identical silicon."""
    fw = meta["F"] * meta["W"]
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
        out = dict(value=pg * fw / sorted(tg)[1] / 1e6, unit="M rays/s",
                   note=f"the oracle's eager PyTorch graph + autograd on this MI355X, {pg * fw} rays of the same workload")
        del gsrc, glv
        torch.cuda.empty_cache()
        return out
    except Exception as e:                      # informational only: never fail the bench on it
        return dict(value=None, note=f"not measured: {type(e).__name__}: {e}"[:200])


def leaf_grad_check(name, device, mode, log2_pupil=18):
    """Gradients of the LEAVES (c, t, nd, v) through the whole RayTracer.trace_rays chain (dispersion, paraxial pupil
    position, assembly, trace, RMS spot), HIP path vs the same chain on the CPU with the oracle's trace in fp32 and
    fp64.  2^log2_pupil pupil points of the workload's fan."""
    import torchoptics_amd as ta
    from oracle import trace_oracle as orc
    from torchoptics_amd import ray_tracing as rt
    dflt = {"cfg3": ((0.707,), ("d",)), "cfg3a": ((0.707,), ("d",)), "cfg3s": ((0.707,), ("d",)), "cfg2": ((0., 0.707, 1.), ("d",)),
            "cfg5": (tuple(np.linspace(0, 1, 5)), ("C", "d", "F"))}[name]
    fields, wl = dflt
    n_r = 1 << (log2_pupil // 2)
    n_th = (1 << log2_pupil) // n_r
    keys = ("c", "t", "nd", "v")

    def gpu():
        lens, specs, leaves = build_lens(name, device)
        tr = ta.RayTracer(mode="circular", n_rays=(n_r, n_th), rel_fields=fields, wavelengths=wl, default_device=device,
                          arith=mode)
        x, y, cx, cy, ok, back = tr.trace_rays(specs, lens)
        rt.compute_rms2d(x, y, ok).backward()
        return {k: leaves[k].grad.detach().cpu() for k in keys}

    def cpu(dtype, ieee):
        from torchoptics_amd import lens_modeling as lm
        lens0, specs0, leaves0 = build_lens(name, "cpu")
        leaves = {k: leaves0[k].detach().to(dtype).requires_grad_(True) for k in leaves0}
        lens = lm.Lens(lens0.structure, leaves["c"], leaves["t"], leaves["nd"], leaves["v"], leaves.get("kappa"),
                       leaves.get("poly"))
        specs = lm.Specs(lens0.structure, specs0.epd.to(dtype), specs0.hfov.to(dtype))
        tr = ta.RayTracer(mode="circular", n_rays=(n_r, n_th), rel_fields=fields, wavelengths=wl, default_device="cpu")
        a = tr.assemble(specs, lens)
        a = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in a.items()}
        if "kappa" in a:
            kap, pol = a["kappa"].reshape(-1), a["poly"].reshape(-1, 4)
            kind = ((kap != 0) | (pol != 0).any(dim=1)).int().tolist()
            x, y, cx, cy, ok, back, _ = orc.trace_skew_general(a["x"], a["y"], a["z"], a["cx"], a["cy"], a["c"], a["t"],
                                                               a["mu"], a["mask"], kap, pol, kind, ieee_sqrt=ieee)
        else:
            x, y, cx, cy, ok, back = orc.trace_skew(a["x"], a["y"], a["z"], a["cx"], a["cy"], a["c"], a["t"], a["mu"],
                                                    a["mask"], ieee_sqrt=ieee)
        orc.compute_rms2d(x, y, ok).backward()
        return {k: leaves[k].grad.detach() for k in keys}

    def rel(a, b):
        return float(((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300)).item())
    try:
        g, r32, r32i, r64 = gpu(), cpu(torch.float32, False), cpu(torch.float32, True), cpu(torch.float64, False)
        live = [k for k in keys if float(r64[k].norm()) > 0.0]       # e.g. d/dv is exactly 0 for a d-line-only fan
        per = {k: dict(vs_fp32_chain_ieee_sqrt=rel(g[k], r32i[k]), vs_fp32_chain_mkl_sqrt=rel(g[k], r32[k]),
                       vs_fp64_chain=rel(g[k], r64[k]), pytorch_fp32_vs_fp64=rel(r32[k], r64[k])) for k in live}
        for k in keys:
            if k not in live:
                per[k] = dict(zero_gradient=True, abs_norm_hip=float(g[k].double().norm()))
        vals = [per[k] for k in live]
        return dict(sample_rays=len(fields) * len(wl) * (1 << log2_pupil), arith_mode=mode, per_leaf=per,
                    max_vs_fp32_chain=max(v["vs_fp32_chain_ieee_sqrt"] for v in vals),
                    max_vs_fp64_chain=max(v["vs_fp64_chain"] for v in vals),
                    max_pytorch_fp32_vs_fp64=max(v["pytorch_fp32_vs_fp64"] for v in vals),
                    note="d(rms)/d(c, t, nd, v) through RayTracer.trace_rays end to end: HIP path (kernels + tl_pupil_position "
                         "+ autograd over the host chain) vs the package's host chain on CPU tensors with the oracle's "
                         "trace_skew / compute_rms2d, in fp32 (both sqrt flavours) and fp64; norm-relative per leaf")
    except Exception as e:                      # a checker problem must not cost the measurement
        return dict(error=f"{type(e).__name__}: {e}"[:300])


if __name__ == "__main__":
    main()
