#!/usr/bin/env python3
"""Time tl_trace_fwd with parts of its output switched off (moments / per-ray outputs) to see what they cost."""
import ctypes as C, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from torchoptics_amd import _lib, ops
mode = sys.argv[1] if len(sys.argv) > 1 else "fast"
dev = torch.device("cuda:0")
args, meta, _ = bench.workload("cfg3", "cuda:0", 1, 0, None)
F, W, P, S = meta["F"], meta["W"], meta["P_local"], meta["S"]
x_e, y_e = args["x"].expand(1, F, P, W), args["y"].expand(1, F, P, W)
prob = ops._problem(x_e, y_e, args["z"].detach().reshape(1).contiguous(), args["cx"].reshape(-1).contiguous(),
                    args["cy"].detach().reshape(-1).contiguous(), args["c"].detach().reshape(S).contiguous(),
                    args["t"].detach().reshape(S).contiguous(), args["mu"].detach().reshape(-1, S).expand(W, S).contiguous(),
                    args["mask"].reshape(-1).to(torch.uint8).contiguous(), True, mode)
outs = [torch.empty((1, F, W, P), dtype=torch.float32, device=dev) for _ in range(4)]
flags = [torch.empty((1, F, W, P), dtype=torch.uint8, device=dev) for _ in range(2)]
mom = torch.empty((F, _lib.TL_NMOM), dtype=torch.float64, device=dev)
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
lib = _lib.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream); P_ = _lib.ptr
variants = {"all": (outs, flags, mom), "no moments": (outs, flags, None), "moments only": ([None]*4, [None]*2, mom),
            "y+ok+moments": ([None, outs[1], None, None], [flags[0], None], mom)}
res = {k: [] for k in variants}
for rnd in range(8):
    for k, (o, f, m) in variants.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.tl_trace_fwd(C.byref(prob), *[P_(t) for t in o], *[P_(t) for t in f], None, None, P_(m), P_(ws), ws.numel(), st)
        e1.record(); torch.cuda.synchronize(); assert rc == 0
        if rnd: res[k].append(e0.elapsed_time(e1))
for k, v in res.items():
    print(f"{mode} fwd [{k:14s}] med {statistics.median(v):.4f} ms")
