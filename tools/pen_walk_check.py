#!/usr/bin/env python3
"""Development check: gradients of rms + 0.2 sumQ (the real caller's loss) through the walk-back and the checkpoint
backward against the oracle's fp64 autograd, per argument group, with and without aspheric rows."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, rel_l2          # noqa: E402
from test_oracle_asphere import asphere_params    # noqa: E402

DEV = "cuda:0"
IN = ("in_x", "in_y", "in_z", "in_cx", "in_cy", "in_c", "in_t", "in_mu")


def main():
    import torchoptics_amd as ta
    from oracle import trace_oracle as orc
    from torchoptics_amd import ops, ray_tracing as rt
    for case in ("G4_tessar_32x32", "G4_doublet_32x32"):
        g = load_golden(case)
        ins = [torch.from_numpy(g[n]) for n in IN]
        mask = torch.from_numpy(g["in_mask"])
        S = ins[5].shape[-1]
        F, P, W = ins[4].shape[1], ins[0].shape[2], ins[7].shape[3]
        x_in, y_in = ins[0].expand(1, F, P, W).contiguous(), ins[1].expand(1, F, P, W).contiguous()
        for asph in (False, True):
            if asph and S < 6:
                continue
            kap0, pol0, kind = asphere_params(S) if asph else (None, None, None)
            names = ("z", "cy", "c", "t", "mu") + (("kappa", "poly") if asph else ())
            base = [ins[2], ins[4], ins[5], ins[6], ins[7]] + ([kap0, pol0] if asph else [])
            res = {}
            for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
                lv = [q.to(dt).clone().requires_grad_(True) for q in base]
                extra = (lv[5], lv[6], kind) if asph else ()
                o = orc.trace_skew_general(x_in.to(dt), y_in.to(dt), lv[0], ins[3].to(dt), lv[1], lv[2], lv[3], lv[4], mask,
                                           *extra, aggregate=True, ieee_sqrt=(dt == torch.float32))
                (orc.compute_rms2d(o[0], o[1], o[4]) + 0.2 * orc.penalty_from_stacks(o[7], S)).backward()
                res[tag] = [q.grad for q in lv]
            for mode in ("strict", "fast"):
                for algo in ("inverse", "checkpoint"):
                    ops.set_backward_algorithm(algo)
                    lv = [q.to(DEV).clone().requires_grad_(True) for q in base]
                    extra = dict(kappa=lv[5], poly=lv[6]) if asph else {}
                    out = ta.trace_skew(x_in.to(DEV), y_in.to(DEV), lv[0], ins[3].to(DEV), lv[1], lv[2], lv[3], lv[4],
                                        mask.to(DEV), aggregate="sum", mode=mode, **extra)
                    (ta.compute_rms2d(out[0], out[1], out[4]) + 0.2 * rt.penalty_sum(out[6], S)).backward()
                    line = f"{case} asph={asph} {mode:6s} {algo:10s} ill={out[1]._tl_spot[0][:, 9].sum().item():.0f}"
                    for n, q, g32, g64 in zip(names, lv, res["f32"], res["f64"]):
                        line += f" | {n} {rel_l2(q.grad.cpu().numpy(), g64.numpy()):.1e} ({rel_l2(g32.numpy(), g64.numpy()):.1e})"
                    print(line)
                ops.set_backward_algorithm("inverse")


if __name__ == "__main__":
    main()
