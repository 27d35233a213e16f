#!/usr/bin/env python3
"""
A/B timing of kernel build variants in ONE process, interleaved rounds (guide rule 24).

    python tools/ab_kernels.py --libs base=torchoptics_amd/libtltrace.so noslp=torchoptics_amd/libtltrace_noslp.so \
        [--workload cfg3] [--mode strict] [--rounds 7] [--log2-pupil 24]

Times tl_trace_fwd and tl_trace_bwd (each including its tiny reduce kernel) with events on the
launch stream; prints median / min per variant.  Development tool, not part of the product.
"""
import argparse
import ctypes as C
import os
import statistics
import sys

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
    a = ap.parse_args()
    import bench
    from torchoptics_amd import _lib, ops
    dev = torch.device("cuda:0")
    args, meta, _ = bench.workload(a.workload, "cuda:0", 1, 0, a.log2_pupil)
    F, W, P, S = meta["F"], meta["W"], meta["P_local"], meta["S"]
    x_e, y_e = args["x"].expand(1, F, P, W), args["y"].expand(1, F, P, W)
    cxv, cyv = args["cx"].reshape(-1).contiguous(), args["cy"].detach().reshape(-1).contiguous()
    mu2 = args["mu"].detach().reshape(-1, S).expand(W, S).contiguous()
    mask = args["mask"].reshape(-1).to(torch.uint8).contiguous()
    prob = ops._problem(x_e, y_e, args["z"].detach().reshape(1).contiguous(), cxv, cyv,
                        args["c"].detach().reshape(S).contiguous(), args["t"].detach().reshape(S).contiguous(),
                        mu2, mask, True, a.mode)
    outs = [torch.empty((1, F, W, P), dtype=torch.float32, device=dev) for _ in range(4)]
    flags = [torch.empty((1, F, W, P), dtype=torch.uint8, device=dev) for _ in range(2)]
    mom = torch.empty((F, _lib.TL_NMOM), dtype=torch.float64, device=dev)
    gmom = torch.randn((F, _lib.TL_NMOM), dtype=torch.float64, device=dev) * 1e-3
    gpar = torch.empty(2 * S + W * S + 1 + 2 * F, dtype=torch.float64, device=dev)   # big enough for either ABI
    g_c, g_t, g_mu, g_z, g_cx, g_cy = torch.split(gpar, [S, S, W * S, 1, F, F])
    ws = torch.zeros(64 << 20, dtype=torch.uint8, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    libs = {}
    import shutil
    import tempfile
    tmpdir = tempfile.mkdtemp(prefix="tl_ab_")
    lib_env = {}
    for spec in a.libs:
        # name=path[@VAR=value,VAR=value]: the launch-plan variables (TL_FWD_BLOCKS, TL_BWD_BLOCKS, ...) are read
        # once per loaded library, so a variant with its own environment gets its own copy of the .so
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
        ver = dll.tl_version()
        for fn, (res, argt) in _lib._SIGNATURES.items():
            if not hasattr(dll, fn):                   # older build: entry point not there yet
                continue
            f = getattr(dll, fn)
            f.restype = res
            if fn == "tl_trace_fwd" and ver < 5:       # older ABI: no `stacks` argument
                argt = argt[:8] + argt[9:]
            if fn == "tl_trace_bwd_from_outputs" and ver < 9:   # older ABI: no g_kappa, g_poly
                argt = argt[:-5] + argt[-3:]
            if fn == "tl_trace_bwd" and ver < 11:               # older ABI: no g_opd, g_n_index
                argt = argt[:-5] + argt[-3:]
            f.argtypes = argt
        dll._ver = ver
        libs[name] = dll
    P_ = _lib.ptr

    def fwd(dll):
        extra = (None, None) if dll._ver >= 5 else (None,)
        rc = dll.tl_trace_fwd(C.byref(prob), *[P_(o) for o in outs], *[P_(f) for f in flags], *extra, P_(mom), P_(ws),
                              ws.numel(), st)
        assert rc == 0, dll.tl_last_error()

    def bwd(dll):
        opd_args = ((None,), (None,)) if dll._ver >= 11 else ((), ())       # ABI 11: g_opd, g_n_index
        rc = dll.tl_trace_bwd(C.byref(prob), None, None, None, None, P_(gmom), *opd_args[0], P_(g_c), P_(g_t), P_(g_mu),
                              P_(g_z), P_(g_cx), P_(g_cy), None, None, *opd_args[1], None, None, P_(ws), ws.numel(), st)
        assert rc == 0, dll.tl_last_error()

    def bwd_inv(dll):
        rc = dll.tl_trace_bwd_from_outputs(C.byref(prob), None, None, None, None, P_(gmom), P_(outs[0]), P_(outs[1]),
                                           P_(outs[2]), P_(outs[3]), P_(flags[0]), P_(mom), P_(g_c), P_(g_t), P_(g_mu),
                                           P_(g_z), P_(g_cx), P_(g_cy), *((None, None) if dll._ver >= 9 else ()),
                                           None, None, P_(ws), ws.numel(), st)
        assert rc == 0, dll.tl_last_error()

    res = {n: {"fwd": [], "bwd": [], "bwd_inv": []} for n in libs}
    ref = {}
    for rnd in range(a.rounds + 1):
        for name, dll in libs.items():
            if rnd == 0:
                for k_, v_ in lib_env.get(name, {}).items():
                    os.environ[k_] = v_
            for key, fn in (("fwd", fwd), ("bwd", bwd), ("bwd_inv", bwd_inv)):
                if key == "bwd_inv" and (dll._ver < 8 or a.workload == "cfg3a"):
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn(dll)
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    res[name][key].append(e0.elapsed_time(e1))
            if rnd == 0:
                ref[name] = (mom.clone(), gpar.clone())
                for k_ in lib_env.get(name, {}):
                    os.environ.pop(k_, None)
    base = next(iter(libs))
    rays = F * W * P
    print(f"workload {a.workload} mode {a.mode}: F={F} W={W} P={P} S={S} ({rays} rays), {a.rounds} rounds")
    for name in libs:
        r = res[name]
        dm = (ref[name][0] - ref[base][0]).abs().max().item()
        dg = ((ref[name][1] - ref[base][1]).norm() / ref[base][1].norm()).item()
        print(f"  {name:14s} fwd med {statistics.median(r['fwd']):.4f} min {min(r['fwd']):.4f} ms | "
              f"bwd med {statistics.median(r['bwd']):.4f} min {min(r['bwd']):.4f} ms | "
              f"fwd+bwd {rays / (statistics.median(r['fwd']) + statistics.median(r['bwd'])) / 1e6:.2f} G rays/s | "
              + (f"walk-back bwd med {statistics.median(r['bwd_inv']):.4f} ms -> "
                 f"{rays / (statistics.median(r['fwd']) + statistics.median(r['bwd_inv'])) / 1e6:.2f} G rays/s | " if r['bwd_inv'] else "") +
              f"d(moments) {dm:.1e} d(grads) {dg:.1e} vs {base}")


if __name__ == "__main__":
    main()
