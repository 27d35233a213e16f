#!/usr/bin/env python3
"""
A/B timing of kernel build variants in ONE process, interleaved rounds (guide rule 24).

    python tools/ab_kernels.py --libs base=torchoptics_amd/libtltrace.so w5=torchoptics_amd/libtltrace_w5.so \
        [--workload cfg3a] [--mode strict] [--rounds 7] [--log2-pupil 24] [--aggregate] [--hit-slots 4]

Times tl_trace_fwd, tl_trace_bwd (checkpoint) and tl_trace_bwd_from_outputs (walk-back), each including its tiny
reduce kernel, with events on the launch stream; prints median / min per variant and how far each variant's moments
and gradients are from the first one's.  A variant may carry its own environment: name=path@VAR=value,VAR=value (the
launch-plan variables are read once per loaded library, so such a variant gets its own copy of the .so).
Development tool, not part of the product (ABI 13 libraries only).
"""
import argparse
import ctypes as C
import os
import shutil
import statistics
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--mode", default="strict")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--log2-pupil", type=int, default=None)
    ap.add_argument("--aggregate", action="store_true", help="penalty term (aggregate='sum')")
    ap.add_argument("--hit-slots", type=int, default=4, help="aspheric hit slots handed to the walk-back (0 = Newton)")
    ap.add_argument("--skip", default="", help="comma list of fwd,bwd,bwd_inv not to time")
    ap.add_argument("--asph-zero", action="store_true",
                    help="all-spherical workload through the ASPHERIC kernel variants: kappa = poly = 0 and no row marked "
                         "aspheric (what the variants cost on spherical rows)")
    ap.add_argument("--cond-flags", action="store_true", help="hand the forward a tl_problem.cond_flags buffer (what the host chains do)")
    a = ap.parse_args()
    import bench
    from torchoptics_amd import _lib, ops
    dev = torch.device("cuda:0")
    args, meta, _ = bench.workload(a.workload, "cuda:0", 1, 0, a.log2_pupil)
    F, W, P, S = meta["F"], meta["W"], meta["P_local"], meta["S"]
    x_e, y_e = args["x"].detach().expand(1, F, P, W), args["y"].detach().expand(1, F, P, W)
    cxv, cyv = args["cx"].reshape(1, -1).contiguous(), args["cy"].detach().reshape(1, -1).contiguous()
    mu2 = args["mu"].detach().reshape(-1, S).expand(W, S).contiguous()
    mask = args["mask"].reshape(-1).to(torch.uint8).contiguous()
    asph = "kappa" in args or a.asph_zero
    kap = pol = kind = hits = None
    if asph:
        kap = args["kappa"].detach().reshape(S).contiguous() if "kappa" in args else torch.zeros(S, device=dev)
        pol = args["poly"].detach().reshape(S, 4).contiguous() if "poly" in args else torch.zeros(S, 4, device=dev)
        kind = ((kap != 0) | (pol != 0).any(dim=1)).to(torch.uint8).contiguous()
        if a.hit_slots:
            hits = torch.empty((a.hit_slots, 2, 1, F, W, P), dtype=torch.float32, device=dev)
    cond = torch.empty((1, F, W, P), dtype=torch.uint8, device=dev) if a.cond_flags else None      # (must outlive prob)
    prob = ops._problem(x_e, y_e, args["z"].detach().reshape(1).contiguous(), cxv, cyv,
                        args["c"].detach().reshape(S).contiguous(), args["t"].detach().reshape(S).contiguous(),
                        mu2, mask, True, a.mode, kap, pol, kind, None, a.aggregate, hits,
                        cond=cond)
    outs = [torch.empty((1, F, W, P), dtype=torch.float32, device=dev) for _ in range(4)]
    flags = [torch.empty((1, F, W, P), dtype=torch.uint8, device=dev) for _ in range(2)]
    mom = torch.empty((F, _lib.TL_NMOM), dtype=torch.float64, device=dev)
    gmom = torch.randn((F, _lib.TL_NMOM), dtype=torch.float64, device=dev) * 1e-3
    gpar = torch.zeros(2 * S + W * S + 1 + 2 * F + 5 * S, dtype=torch.float32, device=dev)
    g_c, g_t, g_mu, g_z, g_cx, g_cy, g_kap, g_pol = torch.split(gpar, [S, S, W * S, 1, F, F, S, 4 * S])
    ws = torch.zeros(256 << 20, dtype=torch.uint8, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    libs, lib_env = {}, {}
    tmpdir = tempfile.mkdtemp(prefix="tl_ab_")
    for spec in a.libs:
        name, path = spec.split("=", 1)
        path, _, envs = path.partition("@")
        src = os.path.join(ROOT, path)
        if envs:
            dst = os.path.join(tmpdir, f"lib_{name}.so")
            shutil.copy(src, dst)
            src = dst
            lib_env[name] = dict(kv.split("=") for kv in envs.split(","))
        dll = C.CDLL(src)
        dll.tl_version.restype = C.c_int
        assert dll.tl_version() == _lib.TL_ABI_VERSION, f"{name}: ABI {dll.tl_version()}"
        for fn, (res, argt) in _lib._SIGNATURES.items():
            f = getattr(dll, fn)
            f.restype, f.argtypes = res, argt
        libs[name] = dll
    P_ = _lib.ptr
    ak, ap_ = (P_(g_kap), P_(g_pol)) if asph else (None, None)

    def fwd(dll):
        rc = dll.tl_trace_fwd(C.byref(prob), *[P_(o) for o in outs], *[P_(f) for f in flags], None, None, P_(mom), P_(ws),
                              ws.numel(), st)
        assert rc == 0, dll.tl_last_error()

    def bwd(dll):
        rc = dll.tl_trace_bwd(C.byref(prob), None, None, None, None, P_(gmom), None, P_(g_c), P_(g_t), P_(g_mu),
                              P_(g_z), P_(g_cx), P_(g_cy), ak, ap_, None, None, None, P_(ws), ws.numel(), st)
        assert rc == 0, dll.tl_last_error()

    def bwd_inv(dll):
        rc = dll.tl_trace_bwd_from_outputs(C.byref(prob), None, None, None, None, P_(gmom), P_(outs[0]), P_(outs[1]),
                                           P_(outs[2]), P_(outs[3]), P_(flags[0]), P_(mom), P_(g_c), P_(g_t), P_(g_mu),
                                           P_(g_z), P_(g_cx), P_(g_cy), ak, ap_, None, None, P_(ws), ws.numel(), st)
        assert rc == 0, dll.tl_last_error()

    skip = set(a.skip.split(",")) if a.skip else set()
    steps = [(k, f) for k, f in (("fwd", fwd), ("bwd", bwd), ("bwd_inv", bwd_inv)) if k not in skip]
    res = {n: {k: [] for k, _ in steps} for n in libs}
    ref = {}
    for rnd in range(a.rounds + 1):
        for name, dll in libs.items():
            if rnd == 0:
                for k_, v_ in lib_env.get(name, {}).items():
                    os.environ[k_] = v_
            for key, fn in steps:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn(dll)
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    res[name][key].append(e0.elapsed_time(e1))
                elif key != "fwd":
                    ref.setdefault(name, {})[key] = gpar.clone()
            if rnd == 0:
                ref.setdefault(name, {})["mom"] = mom.clone()
                for k_ in lib_env.get(name, {}):
                    os.environ.pop(k_, None)
    base = next(iter(libs))
    rays = F * W * P
    rel = lambda x, y: ((x.double() - y.double()).norm() / y.double().norm().clamp_min(1e-300)).item()      # noqa: E731
    print(f"workload {a.workload} mode {a.mode} aggregate {a.aggregate}: F={F} W={W} P={P} S={S} ({rays} rays), "
          f"hit slots {a.hit_slots if asph else '-'}, {a.rounds} rounds")
    for name in libs:
        r = res[name]
        med = {k: statistics.median(v) for k, v in r.items()}
        line = f"  {name:12s}" + "".join(f" {k} med {med[k]:.4f} min {min(r[k]):.4f} ms |" for k in med)
        if "fwd" in med and "bwd_inv" in med:
            line += f" fwd+walk-back {rays / (med['fwd'] + med['bwd_inv']) / 1e6:.2f} G rays/s |"
        line += f" d(moments) {(ref[name]['mom'] - ref[base]['mom']).abs().max().item():.1e}"
        for k in ("bwd", "bwd_inv"):
            if k in ref[name]:
                line += f" d({k} grads) {rel(ref[name][k], ref[base][k]):.1e}"
        if "bwd" in ref[name] and "bwd_inv" in ref[name]:
            line += f" | walk-back vs checkpoint {rel(ref[name]['bwd_inv'], ref[name]['bwd']):.1e}"
        print(line)


if __name__ == "__main__":
    main()
