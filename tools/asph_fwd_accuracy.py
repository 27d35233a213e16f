#!/usr/bin/env python3
"""Forward accuracy of two library builds on the aspheric bench workloads against the double-precision kernels: rms, per-ray
image positions, masks.  base = torchoptics_amd/libtltrace_base.so (a copy of the build to compare with), new = libtltrace.so.
Development tool."""
import ctypes as C, os, sys
import torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from torchoptics_amd import _lib, ops
for wl in ("cfg3a", "cfg3s"):
    args, meta, _ = bench.workload(wl, "cuda:0", 1, 0, None)
    F, W, P, S = meta["F"], meta["W"], meta["P_local"], meta["S"]
    dev = torch.device("cuda:0")
    x_e, y_e = args["x"].detach().expand(1, F, P, W), args["y"].detach().expand(1, F, P, W)
    keep = [args["z"].detach().reshape(1).contiguous(), args["cx"].reshape(1, -1).contiguous(), args["cy"].detach().reshape(1, -1).contiguous(),
            args["c"].detach().reshape(S).contiguous(), args["t"].detach().reshape(S).contiguous(),
            args["mu"].detach().reshape(-1, S).expand(W, S).contiguous(), args["mask"].reshape(-1).to(torch.uint8).contiguous()]
    kap, pol = args["kappa"].detach().reshape(S).contiguous(), args["poly"].detach().reshape(S, 4).contiguous()
    kind = ((kap != 0) | (pol != 0).any(dim=1)).to(torch.uint8).contiguous()
    prob = ops._problem(x_e, y_e, *keep, True, "strict", kap, pol, kind)
    ws = torch.zeros(256 << 20, dtype=torch.uint8, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    res = {}
    for name in ("base", "new"):
        dll = C.CDLL(os.path.join(ROOT, "torchoptics_amd", "libtltrace_base.so" if name == "base" else "libtltrace.so"))
        for fn, (r_, a_) in _lib._SIGNATURES.items():
            f = getattr(dll, fn); f.restype, f.argtypes = r_, a_
        mom = torch.empty((F, _lib.TL_NMOM), dtype=torch.float64, device=dev)
        outs = [torch.empty((1, F, W, P), dtype=torch.float32, device=dev) for _ in range(4)]
        fl = [torch.empty((1, F, W, P), dtype=torch.uint8, device=dev) for _ in range(2)]
        assert dll.tl_trace_fwd(C.byref(prob), *[_lib.ptr(o) for o in outs], *[_lib.ptr(f) for f in fl], None, None, _lib.ptr(mom), _lib.ptr(ws), ws.numel(), st) == 0
        torch.cuda.synchronize()
        M = mom[0].cpu(); n = P * W
        m = M[0] / n
        rms = float(torch.sqrt((M[2] - 2 * m * M[1] + m * m * M[3]) / n))
        res[name] = (rms, outs[0].clone(), outs[1].clone(), fl[0].clone())
    # fp64 kernels
    a64 = [t.double() for t in (x_e.contiguous(), y_e.contiguous())]
    k64 = [k.double() if k.dtype == torch.float32 else k for k in keep]
    prob64 = ops._problem(a64[0], a64[1], *k64, True, "strict", kap.double(), pol.double(), kind)
    mom64 = torch.empty((F, _lib.TL_NMOM), dtype=torch.float64, device=dev)
    o64 = [torch.empty((1, F, W, P), dtype=torch.float64, device=dev) for _ in range(4)]
    f64 = [torch.empty((1, F, W, P), dtype=torch.uint8, device=dev) for _ in range(2)]
    lib = _lib.lib()
    assert lib.tl_trace_fwd_f64(C.byref(prob64), *[_lib.ptr(o) for o in o64], *[_lib.ptr(f) for f in f64], _lib.ptr(mom64), _lib.ptr(ws), ws.numel(), st) == 0
    torch.cuda.synchronize()
    M = mom64[0].cpu(); m = M[0] / n
    rms64 = float(torch.sqrt((M[2] - 2 * m * M[1] + m * m * M[3]) / n))
    for name in ("base", "new"):
        rms, x, y, ok = res[name]
        live = (ok != 0) & (f64[0] != 0)
        dx = (x.double() - o64[0])[live].abs(); dy = (y.double() - o64[1])[live].abs()
        print(f"{wl} {name}: rms {rms:.10f} (fp64 {rms64:.10f}, rel {abs(rms - rms64) / rms64:.2e}); |x - x64| max {dx.max():.2e} mean {dx.mean():.2e}; |y - y64| max {dy.max():.2e} mean {dy.mean():.2e}; ok mismatch {int(((ok != 0) != (f64[0] != 0)).sum())}")
