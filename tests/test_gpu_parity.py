"""
GPU parity tests (run on the MI355X box with -m gpu).  Every test calls the HIP kernels
through the C ABI (via torchoptics_amd) and checks them against the golden fixtures produced
by the reference and/or the CPU oracle on the same inputs.

Tolerances
  forward, strict mode : BIT-EXACT (x, y, cx, cy, ok, back) vs the oracle evaluated with a correctly
                         rounded sqrt (oracle ieee_sqrt=True).  The reference's own CPU sqrt (MKL VML) is
                         up to 1 ulp off and tensor-size dependent, so vs the reference fixtures: masks
                         identical, positions within 1e-5 mm, cosines within 5e-7 (same bound the IEEE
                         oracle itself meets, tests/test_oracle_golden.py)
  rms                  : |d| <= 2e-7 relative (moments are accumulated in fp64, the reference sums in fp32)
  gradients, strict    : norm-relative error <= 1e-5 vs the reference's fp32 autograd
                         AND <= 3e-5 vs its fp64 autograd (the fp32 reference itself is 2e-5 away
                         from fp64)
  fast mode            : forward within 2e-5 absolute [mm], gradients <= 1e-4 norm-relative
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu


def _used_walk_back(t):
    from torchoptics_amd import ops
    return ops.used_walk_back(t)

RAY_CASES = ["G1_singlet_cfg1", "G2_cooke_16x16", "G4_doublet_32x32", "G4_tessar_32x32",
             "G5_cooke_failures", "G6_cooke_aim1", "G10_cooke_noback", "G10_tessar_noback"]
IN_NAMES = ("in_x", "in_y", "in_z", "in_cx", "in_cy", "in_c", "in_t", "in_mu")
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ta():
    import torchoptics_amd
    from torchoptics_amd import _lib
    _lib.lib()        # the HIP library must be present: no fallback exists
    return torchoptics_amd


def dev_inputs(g, grad=False):
    ins = [torch.from_numpy(g[n]).to(DEV).requires_grad_(grad) for n in IN_NAMES]
    return ins, torch.from_numpy(g["in_mask"]).to(DEV), bool(g.get("allow_backward_rays", True))


@pytest.mark.parametrize("case", RAY_CASES)
def test_forward_strict_bit_exact_vs_ieee_oracle_and_close_to_reference(ta, case):
    from oracle import trace_oracle as orc
    g = load_golden(case)
    ins, mask, allow = dev_inputs(g)
    x, y, cx, cy, ok, back = ta.trace_skew(*ins, mask, False, allow, mode="strict")
    assert x.shape == g["x"].shape and ok.dtype == torch.bool
    cpu = [torch.from_numpy(g[n]) for n in IN_NAMES]
    want = orc.trace_skew(*cpu, torch.from_numpy(g["in_mask"]), False, allow, ieee_sqrt=True)
    for name, got, ref in zip(("x", "y", "cx", "cy", "ok", "back"), (x, y, cx, cy, ok, back), want):
        # with allow_backward_rays=False the reference never updates ray_backward, which then
        # keeps the un-broadcast shape of its y argument (all False): compare broadcast
        got = got.cpu().numpy()
        ref = np.broadcast_to(ref.numpy(), got.shape)
        if not np.array_equal(got, ref):
            bad = np.flatnonzero(got.ravel() != ref.ravel())
            pytest.fail(f"{case}:{name} differs from the IEEE oracle at {bad.size}/{got.size} rays")
    # versus the reference's own (MKL-sqrt) outputs
    assert np.array_equal(ok.cpu().numpy(), g["ok"])
    assert np.array_equal(back.cpu().numpy(), np.broadcast_to(g["back"], back.shape))
    for name, got, tol in (("x", x, 1e-5), ("y", y, 1e-5), ("cx", cx, 5e-7), ("cy", cy, 5e-7)):
        assert np.abs(got.cpu().numpy() - g[name]).max() <= tol, name
    rms = ta.compute_rms2d(x, y, ok)
    # our moments are summed in fp64; the reference sums in fp32 (its own rms is ~2e-6 off its fp64 value)
    assert abs(rms.item() - float(g["rms_in64"])) <= 2e-6 * abs(float(g["rms_in64"])) + 1e-9
    assert abs(rms.item() - float(g["rms_in"])) <= 5e-6 * abs(float(g["rms_in"])) + 1e-9


@pytest.mark.parametrize("case", RAY_CASES)
def test_backward_vs_reference_autograd(ta, case):
    g = load_golden(case)
    ins, mask, allow = dev_inputs(g, grad=True)
    x, y, cx, cy, ok, back = ta.trace_skew(*ins, mask, False, allow, mode="strict")
    rms = ta.compute_rms2d(x, y, ok)
    grads = torch.autograd.grad(rms, ins, allow_unused=True)
    report = {}
    for n, got in zip(("x", "y", "z", "cx", "cy", "c", "t", "mu"), grads):
        w32, w64 = g["gin_" + n], g["gin_" + n + "64"]
        got = np.zeros_like(w32) if got is None else got.cpu().numpy()
        assert got.shape == w32.shape, n
        if np.linalg.norm(w64) < 1e-12:       # identically-zero gradient (e.g. d/dcx on axis)
            assert np.abs(got).max() < 1e-6
            continue
        report[n] = (rel_l2(got, w32), rel_l2(got, w64), rel_l2(w32, w64))
    print(case, {k: tuple(f"{e:.1e}" for e in v) for k, v in report.items()})
    for n, (e32, e64, ref_noise) in report.items():
        if n == "cx":      # d/dcx is pure rounding noise for meridional fans (value ~1e-10)
            continue
        # lens parameters c, t, mu: 1e-5.  Launch conditions: per-ray d/dx, d/dy are single fp32
        # chains with no averaging, and d/dz, d/dcy are small residuals of large per-ray terms
        # (the seeds are mean-free and d y/d z is nearly the same for every ray): 3e-5, or within the
        # fp32 reference's own distance from fp64.
        tol = 1e-5 if n in ("c", "t", "mu") else 3e-5
        assert e32 <= tol or e64 <= max(tol, ref_noise), f"{case} d/d{n}: vs fp32 {e32:.2e}, vs fp64 {e64:.2e}"
        assert e64 <= 3e-5 + ref_noise, f"{case} d/d{n}: vs fp64 {e64:.2e}"


def test_dense_upstream_gradients_match_oracle(ta):
    """Arbitrary loss on all four outputs (dense gx, gy, gcx, gcy path of the backward kernel)."""
    from oracle import trace_oracle as orc
    g = load_golden("G5_cooke_failures")
    ins, mask, allow = dev_inputs(g, grad=True)
    gen = torch.Generator().manual_seed(1)
    wts = [torch.randn(g["x"].shape, generator=gen) for _ in range(4)]
    outs = ta.trace_skew(*ins, mask, False, allow, mode="strict")
    loss = sum((o * w.to(DEV)).sum() for o, w in zip(outs[:4], wts))
    got = torch.autograd.grad(loss, ins)
    want = {}
    for dt in (torch.float32, torch.float64):
        cpu = [torch.from_numpy(g[n]).to(dt).requires_grad_(True) for n in IN_NAMES]
        ref_outs = orc.trace_skew(*cpu, torch.from_numpy(g["in_mask"]), False, allow, ieee_sqrt=True)
        ref_loss = sum((o * w.to(dt)).sum() for o, w in zip(ref_outs[:4], wts))
        want[dt] = torch.autograd.grad(ref_loss, cpu)
    for n, a, b32, b64 in zip(IN_NAMES, got, want[torch.float32], want[torch.float64]):
        e32, e64 = rel_l2(a.cpu().numpy(), b32.numpy()), rel_l2(a.cpu().numpy(), b64.numpy())
        noise = rel_l2(b32.numpy(), b64.numpy())      # what fp32 autograd itself loses
        assert e32 < 1e-5 or e64 < 2 * noise + 1e-6, f"{n}: vs fp32 {e32:.2e}, vs fp64 {e64:.2e}, fp32 noise {noise:.2e}"


def test_generic_spot_path_equals_fused(ta):
    """compute_rms2d on tensors that lost the fused tag goes through tl_spot_moments/tl_spot_seed."""
    g = load_golden("G5_cooke_failures")
    ins, mask, allow = dev_inputs(g, grad=True)
    x, y, cx, cy, ok, back = ta.trace_skew(*ins, mask, False, allow)
    fused = ta.compute_rms2d(x, y, ok)
    generic = ta.compute_rms2d(x, y * 1.0, ok)
    assert abs(fused.item() - generic.item()) < 1e-9
    ga = torch.autograd.grad(fused, ins[5:], retain_graph=True)
    gb = torch.autograd.grad(generic, ins[5:])
    for a, b in zip(ga, gb):
        assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("P", [1, 63, 64, 257, 1000, 4097])
def test_ragged_sizes_match_oracle(ta, P):
    """Pupil counts that are not multiples of the wave / block size."""
    from oracle import trace_oracle as orc
    g = load_golden("G4_tessar_32x32")
    gen = torch.Generator().manual_seed(P)
    ins_cpu = [torch.from_numpy(g[n]) for n in IN_NAMES]
    ins_cpu[0] = (torch.rand(1, 1, P, 1, generator=gen) - 0.5) * 8
    ins_cpu[1] = (torch.rand(1, 1, P, 1, generator=gen) - 0.5) * 8
    mask = torch.from_numpy(g["in_mask"])
    want = orc.trace_skew(*ins_cpu, mask, ieee_sqrt=True)
    got = ta.trace_skew(*[a.to(DEV) for a in ins_cpu], mask.to(DEV), mode="strict")
    for a, b in zip(got, want):
        assert torch.equal(a.cpu(), b)
    assert abs(ta.compute_rms2d(*[got[i] for i in (0, 1, 4)]).item() - orc.compute_rms2d(want[0], want[1], want[4]).item()) < 1e-6


def test_multi_round_launch_with_ragged_tail(ta):
    """5.3 M pupil points: every forward block traces 3 rounds (the fast-mode kernel 2 rays per lane per round,
    so one full and one half round), the backward kernels more, and the last block of each is partial.
    strict forward bit-equal to the oracle, fast close to strict, walk-back gradients equal to the
    checkpoint kernel's to rounding."""
    from oracle import trace_oracle as orc
    from torchoptics_amd import ops
    P = 5_300_003
    g = load_golden("G4_tessar_32x32")
    gen = torch.Generator().manual_seed(11)
    ins_cpu = [torch.from_numpy(g[n]) for n in IN_NAMES]
    ins_cpu[0] = (torch.rand(1, 1, P, 1, generator=gen) - 0.5) * 9        # a few percent of the fan misses
    ins_cpu[1] = (torch.rand(1, 1, P, 1, generator=gen) - 0.5) * 9
    mask = torch.from_numpy(g["in_mask"])
    want = orc.trace_skew(*ins_cpu, mask, ieee_sqrt=True)
    dev = [a.to(DEV) for a in ins_cpu]
    got = ta.trace_skew(*dev, mask.to(DEV), mode="strict")
    for a, b in zip(got, want):
        assert torch.equal(a.cpu(), b)
    assert 0.5 < want[4].float().mean().item() < 0.999
    fast = ta.trace_skew(*dev, mask.to(DEV), mode="fast")
    same = (fast[4] == got[4])
    assert same.float().mean().item() > 0.9999                            # flags flip only at a threshold
    for a, b in zip(fast[:4], got[:4]):
        d = (a - b)[same].abs()                                              # near-grazing rays amplify the
        assert (d < 5e-5).float().mean().item() > 0.9999 and d.max().item() < 1e-2   # rounding difference by 1/cos^2
    rms_s = ta.compute_rms2d(got[0], got[1], got[4]).item()
    assert abs(rms_s - orc.compute_rms2d(want[0], want[1], want[4]).item()) < 1e-6
    assert abs(ta.compute_rms2d(fast[0], fast[1], fast[4]).item() - rms_s) < 1e-5
    for mode in ("strict", "fast"):
        grads = {}
        for algo in ("inverse", "checkpoint"):
            ops.set_backward_algorithm(algo)
            try:
                lv = [dev[i].clone().requires_grad_(True) for i in (5, 6, 7)]
                x, y, cx, cy, ok, back = ta.trace_skew(*dev[:5], *lv, mask.to(DEV), mode=mode)
                assert _used_walk_back(x) is (algo == "inverse")
                ta.compute_rms2d(x, y, ok).backward()
                grads[algo] = [q.grad.cpu().numpy() for q in lv]
            finally:
                ops.set_backward_algorithm("inverse")
        for a, b in zip(grads["inverse"], grads["checkpoint"]):
            assert rel_l2(a, b) < (2e-5 if mode == "strict" else 2e-4)


def test_all_rays_fail_gives_zeros_and_zero_grads(ta):
    g = load_golden("G2_cooke_16x16")
    ins, mask, allow = dev_inputs(g, grad=True)
    full = (1, 3, 256, 3)
    big = [(ins[0].detach() * 100).expand(full).contiguous().requires_grad_(True),
           (ins[1].detach() * 100).expand(full).contiguous().requires_grad_(True)] + ins[2:]
    x, y, cx, cy, ok, back = ta.trace_skew(*big, mask)
    sel = ~ok
    assert sel.float().mean().item() > 0.9
    assert x[sel].abs().max().item() == 0 and y[sel].abs().max().item() == 0
    assert cx[sel].abs().max().item() == 0 and cy[sel].abs().max().item() == 0
    grads = torch.autograd.grad((x * x + y * y + cx + cy).sum(), big)
    assert all(torch.isfinite(gr).all() for gr in grads)
    assert grads[0][sel].abs().max().item() == 0 and grads[1][sel].abs().max().item() == 0


def test_fast_mode_close_to_strict(ta):
    g = load_golden("G4_tessar_32x32")
    ins, mask, allow = dev_inputs(g, grad=True)
    res = {}
    for mode in ("strict", "fast"):
        x, y, cx, cy, ok, back = ta.trace_skew(*ins, mask, False, allow, mode=mode)
        rms = ta.compute_rms2d(x, y, ok)
        res[mode] = (x, y, ok, rms, torch.autograd.grad(rms, ins[5:]))
    assert torch.equal(res["strict"][2], res["fast"][2])
    assert (res["strict"][1] - res["fast"][1]).abs().max().item() < 2e-5
    assert abs(res["strict"][3].item() - res["fast"][3].item()) < 1e-5 * res["strict"][3].item()
    for a, b in zip(res["strict"][4], res["fast"][4]):
        assert rel_l2(b.cpu().numpy(), a.cpu().numpy()) < 1e-4


def test_full_chain_cooke_vs_reference_leaf_grads(ta):
    """RayTracer.trace_rays on the GPU from the YAML-equivalent leaves: rms and d/d(c,t,nd,v)."""
    import yaml_free_lenses as L
    g = load_golden("G2_cooke_16x16")
    lens, specs, leaves = L.build("cooke", DEV)
    tr = ta.RayTracer(mode="circular", n_rays=(16, 16), rel_fields=(0., 0.707, 1.), wavelengths=("C", "d", "F"),
                      default_device=DEV)
    x, y, cx, cy, ok, back = tr.trace_rays(specs, lens)
    rms = ta.compute_rms2d(x, y, ok)
    assert abs(rms.item() - float(g["rms"])) < 5e-7
    grads = torch.autograd.grad(rms, [leaves[k] for k in ("c", "t", "nd", "v")])
    for k, got in zip(("c", "t", "nd"), grads):
        e32, e64 = rel_l2(got.cpu().numpy(), g["g_" + k]), rel_l2(got.cpu().numpy(), g["g_" + k + "64"])
        print(f"full chain d/d{k}: vs fp32 autograd {e32:.2e}, vs fp64 autograd {e64:.2e}")
        assert e32 < 1e-5, f"d/d{k}: vs fp32 {e32:.2e} vs fp64 {e64:.2e}"      # north_star: <= 1e-5 of PyTorch autograd


@pytest.mark.parametrize("case,name,n_rays,wl,epd,hfov", [
    ("G1_singlet_cfg1", "singlet", (64, 64), ("d",), None, 25.0),
    ("G2_cooke_16x16", "cooke", (16, 16), ("C", "d", "F"), None, 25.0),
    ("G4_doublet_32x32", "doublet", (32, 32), ("C", "d", "F"), None, 25.0),
    ("G4_tessar_32x32", "tessar", (32, 32), ("C", "d", "F"), None, 25.0),
    ("G5_cooke_failures", "cooke", (32, 32), ("C", "d", "F"), 16.0, 35.0),
])
def test_strict_full_chain_reproduces_the_reference_kernel_inputs_and_outputs(ta, case, name, n_rays, wl, epd, hfov):
    """Strict mode through RayTracer on the GPU: the launch conditions the host chain hands to the kernel are the
    REFERENCE's fp32 values bit for bit -- z from its pairwise fp32 ABCD tree (tl_pupil_position, mode strict), cy from
    the correctly rounded sine, c, t, mu, mask as they are; the fan x, y from correctly rounded cos / sin (the
    reference's own grid is its host libm's, 1 ulp off on a few points) -- and so the per-ray outputs are the IEEE
    oracle's on the reference's inputs wherever the fan agrees."""
    import yaml_free_lenses as L
    from oracle import trace_oracle as orc
    g = load_golden(case)
    lens, specs, _ = L.build(name, DEV, epd=epd or L.EPD, hfov_deg=hfov, grad=False)
    fields = (0.,) if name == "singlet" else (0., 0.707, 1.)
    tr = ta.RayTracer(mode="circular", n_rays=n_rays, rel_fields=fields, wavelengths=wl, default_device=DEV, arith="strict")
    a = tr.assemble(specs, lens)
    for k in ("z", "cy", "cx", "c", "t", "mu", "mask"):
        assert np.array_equal(a[k].cpu().numpy(), g["in_" + k]), f"{case}: in_{k} is not the reference's value"
    for k in ("x", "y"):
        got, want = a[k].cpu().numpy(), g["in_" + k]
        ulp = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
        near_zero = np.abs(want) < 1e-6                      # cos(pi/2) and friends: tiny values, many ulps, ~1e-8 mm apart
        # (the fixture machine's fp32 sin / cos is the correctly rounded value on 81-97 % of these grid angles)
        assert ulp[~near_zero].max() <= 2 and np.abs(got - want)[near_zero].max(initial=0.0) < 1e-6
        assert (ulp == 0).mean() > 0.75, f"{case}: only {(ulp == 0).mean():.3f} of in_{k} bit-equal"
    out = ta.trace_skew(a["x"], a["y"], a["z"], a["cx"], a["cy"], a["c"], a["t"], a["mu"], a["mask"], mode="strict")
    cpu = [a[k].cpu() for k in ("x", "y", "z", "cx", "cy", "c", "t", "mu", "mask")]
    want = orc.trace_skew(*cpu, ieee_sqrt=True)
    for nme, u, v in zip(("x", "y", "cx", "cy", "ok", "back"), out, want):
        assert torch.equal(u.cpu(), v), f"{case}: {nme} differs from the IEEE oracle on the same inputs"


def test_cfg2_full_size_scalars(ta):
    """cfg2: Cooke, 1024x1024 pupil x 3 fields (3.1 M rays) against the reference's scalars."""
    import yaml_free_lenses as L
    g = load_golden("G3_cooke_cfg2_d")
    lens, specs, leaves = L.build("cooke", DEV)
    tr = ta.RayTracer(mode="circular", n_rays=(1024, 1024), rel_fields=(0., 0.707, 1.), wavelengths=("d",),
                      default_device=DEV)
    x, y, cx, cy, ok, back = tr.trace_rays(specs, lens)
    rms = ta.compute_rms2d(x, y, ok)
    assert ok.all().item()
    assert abs(back.float().mean().item() - float(g["back_frac"])) < 1e-6
    # rms_in64 = the reference evaluated in fp64 on the same fp32-valued kernel inputs
    assert abs(rms.item() - float(g["rms_in64"])) < 1e-5 * float(g["rms_in64"])
    assert abs(rms.item() - float(g["rms"])) < 1e-5 * float(g["rms"])
    grads = torch.autograd.grad(rms, [leaves[k] for k in ("c", "t", "nd")])
    for k, got in zip(("c", "t", "nd"), grads):
        e32, e64 = rel_l2(got.cpu().numpy(), g["g_" + k]), rel_l2(got.cpu().numpy(), g["g_" + k + "64"])
        ref = rel_l2(g['g_' + k], g['g_' + k + '64'])
        print(f"cfg2 d/d{k}: vs fp32 autograd {e32:.2e}, vs fp64 autograd {e64:.2e}, fp32-vs-fp64 of the reference {ref:.2e}")
        # <= 1e-5 of the reference's fp32 autograd -- or at least as close to fp64 as the reference's own fp32 run is
        # (its 1024 x 1024 grid comes from its host libm's cos / sin, ours from correctly rounded ones: two fp32
        # evaluations of the same fan)
        assert e32 < 1e-5 or e64 <= ref, f"d/d{k}: vs fp32 {e32:.2e}, vs fp64 {e64:.2e} (reference fp32 itself {ref:.2e})"


@pytest.mark.parametrize("fused", [True, False])
def test_two_dimensional_rms_extension(ta, fused):
    """compute_rms_spot_xy (x- and y-moments from the kernel when the trace was asked for them, a second pass over the
    rays otherwise) vs the same statistic formed with plain torch ops on the per-ray outputs, and its gradient vs
    autograd through the dense-gradient path."""
    from torchoptics_amd import ray_tracing as rt
    g = load_golden("G5_cooke_failures")
    ins, mask, allow = dev_inputs(g, grad=True)
    x, y, cx, cy, ok, back = ta.trace_skew(*ins, mask, False, allow, x_moments=fused)
    mom = y._tl_spot[0]
    assert (mom[:, 4:7].abs().sum().item() > 0) is fused          # x-moments only when asked for
    got = rt.compute_rms_spot_xy(x, y, ok)
    n = y.shape[2] * y.shape[3]
    okd, xd, yd = ok[0].double(), x[0].double(), y[0].double()
    mx, my = xd.sum(dim=(1, 2), keepdim=True) / n, yd.sum(dim=(1, 2), keepdim=True) / n
    want = torch.sqrt((okd * ((xd - mx) ** 2 + (yd - my) ** 2)).sum(dim=(1, 2)) / n).mean()
    assert abs(got.item() - want.item()) <= 1e-6 * want.item()
    ga = torch.autograd.grad(got, ins[5:], retain_graph=True)
    gb = torch.autograd.grad(want, ins[5:])
    for a, b in zip(ga, gb):
        assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 2e-5


@pytest.mark.parametrize("case", [c for c in RAY_CASES if "noback" not in c])
@pytest.mark.parametrize("mode", ["strict", "fast"])
def test_checkpoint_free_backward_vs_reference_autograd(ta, case, mode):
    """tl_trace_bwd_from_outputs (walk back from the forward's outputs, no checkpoints): gradients of the
    lens parameters vs the reference's autograd, and vs the checkpoint kernel, on every fixture incl. the
    failure-heavy one.  strict: 1e-5 vs fp32 autograd (or within the fp32 reference's own distance from
    fp64); fast: 1e-4.  Launch conditions z, cy (residuals ~1e-3 of their per-ray terms): within
    max(3 x the reference's own fp32-vs-fp64 distance, 3e-5) of the fp64 gradient (x10 in fast mode)."""
    from torchoptics_amd import ops
    g = load_golden(case)
    res = {}
    for algo in ("inverse", "checkpoint"):
        ops.set_backward_algorithm(algo)
        try:
            ins, mask, allow = dev_inputs(g)
            lv = [ins[i].clone().requires_grad_(True) for i in (2, 4, 5, 6, 7)]
            x, y, cx, cy, ok, back = ta.trace_skew(ins[0], ins[1], lv[0], ins[3], lv[1], lv[2], lv[3], lv[4], mask,
                                                   False, allow, mode=mode)
            ctx_inv = _used_walk_back(x)
            ta.compute_rms2d(x, y, ok).backward()
            res[algo] = ([q.grad.cpu().numpy() for q in lv], ctx_inv)
        finally:
            ops.set_backward_algorithm("inverse")
    assert res["inverse"][1] is True and res["checkpoint"][1] is False       # each ran the intended kernel
    tol = 1e-5 if mode == "strict" else 1e-4
    for n, gi, gc_ in zip(("z", "cy", "c", "t", "mu"), *[res[k][0] for k in ("inverse", "checkpoint")]):
        w32, w64 = g["gin_" + n], g["gin_" + n + "64"]
        if np.linalg.norm(w64) < 1e-6 * np.linalg.norm(g["gin_c64"]):     # zero by symmetry (on-axis-only fans)
            continue
        e32, e64, noise = rel_l2(gi, w32), rel_l2(gi, w64), rel_l2(w32, w64)
        print(f"{case} {mode} d/d{n}: vs fp32 {e32:.2e} vs fp64 {e64:.2e} (reference fp32 vs fp64 {noise:.2e}) "
              f"walk-back vs checkpoint {rel_l2(gi, gc_):.2e}")
        if n in ("c", "t", "mu"):
            lim, lim_ab = tol, 2e-5
        else:
            lim = max(3 * noise, 3e-5) * (1 if mode == "strict" else 10)
            lim_ab = 2 * lim
        assert e32 <= lim or e64 <= max(lim, 2 * noise), f"{case} {mode} d/d{n}: vs fp32 {e32:.2e}, vs fp64 {e64:.2e}"
        assert rel_l2(gi, gc_) <= lim_ab, f"{case} {mode} d/d{n}: inverse vs checkpoint"


def test_checkpoint_free_backward_with_dense_upstream_gradients(ta):
    """Arbitrary loss on x, y, cx, cy (dense seeds) through the walk-back kernel vs fp64 oracle autograd."""
    from oracle import trace_oracle as orc
    g = load_golden("G4_tessar_32x32")
    ins, mask, allow = dev_inputs(g)
    lv = [ins[i].clone().requires_grad_(True) for i in (5, 6, 7)]
    gen = torch.Generator().manual_seed(3)
    wts = [torch.randn(g["x"].shape, generator=gen) for _ in range(4)]
    outs = ta.trace_skew(*ins[:5], *lv, mask, False, allow)
    assert _used_walk_back(outs[0]) is True
    loss = sum((o * w.to(DEV)).sum() for o, w in zip(outs[:4], wts))
    got = torch.autograd.grad(loss, lv)
    cpu = [torch.from_numpy(g[n]).double() for n in IN_NAMES]
    clv = [cpu[i].clone().requires_grad_(True) for i in (5, 6, 7)]
    ref_outs = orc.trace_skew(*cpu[:5], *clv, torch.from_numpy(g["in_mask"]), False, allow)
    want = torch.autograd.grad(sum((o * w.double()).sum() for o, w in zip(ref_outs[:4], wts)), clv)
    for n, a, b in zip(("c", "t", "mu"), got, want):
        assert rel_l2(a.cpu().numpy(), b.numpy()) < 2e-5, n


def test_conditioning_count_selects_the_backward_kernel_on_the_device(ta):
    """Moment 9 = number of live rays with min cos^2 < 0.01 along their path.  Zero for every sane fan (the
    walk-back kernel then does the backward); non-zero for the failure-heavy fixture, where the checkpoint
    kernel takes those rays inside the same call (the forward flags them per ray: test_gpu_conditioning.py), so that
    gradients stay within 1e-5 of autograd there too."""
    counts = {}
    for case in ("G2_cooke_16x16", "G4_tessar_32x32", "G5_cooke_failures"):
        g = load_golden(case)
        ins, mask, allow = dev_inputs(g)
        x, y, cx, cy, ok, back = ta.trace_skew(*ins, mask, False, allow)
        counts[case] = y._tl_spot[0][:, 9].sum().item()
    assert counts["G2_cooke_16x16"] == 0 and counts["G4_tessar_32x32"] == 0
    assert counts["G5_cooke_failures"] > 0
    # and the split result is the checkpoint algorithm's up to the walk-back's rounding on the well-conditioned rays
    from torchoptics_amd import ops
    g = load_golden("G5_cooke_failures")
    grads = {}
    for algo in ("inverse", "checkpoint"):
        ops.set_backward_algorithm(algo)
        try:
            ins, mask, allow = dev_inputs(g)
            lv = [ins[i].clone().requires_grad_(True) for i in (5, 6, 7)]
            x, y, cx, cy, ok, back = ta.trace_skew(*ins[:5], *lv, mask, False, allow)
            ta.compute_rms2d(x, y, ok).backward()
            grads[algo] = [q.grad.clone() for q in lv]
        finally:
            ops.set_backward_algorithm("inverse")
    for a, b in zip(grads["inverse"], grads["checkpoint"]):
        assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-5


def test_walk_back_non_finite_adjoint_falls_back_to_checkpoint_kernel(ta):
    """The walk-back kernel carries no per-surface mask: a non-finite adjoint (which a well-conditioned kept
    ray cannot produce) raises the poison word in the workspace instead, and the checkpoint kernel queued
    behind it redoes the launch from the inputs.  Provoked here by corrupting the saved forward outputs of
    one live ray between forward and backward: the gradients must be bit-equal to the checkpoint algorithm's."""
    from torchoptics_amd import ops
    g = load_golden("G4_tessar_32x32")
    grads = {}
    for algo in ("inverse", "checkpoint"):
        ops.set_backward_algorithm(algo)
        try:
            ins, mask, allow = dev_inputs(g)
            lv = [ins[i].clone().requires_grad_(True) for i in (5, 6, 7)]
            x, y, cx, cy, ok, back = ta.trace_skew(*ins[:5], *lv, mask, False, allow)
            assert _used_walk_back(x) is (algo == "inverse")
            loss = ta.compute_rms2d(x, y, ok)                  # from the fused moments: untouched by the corruption
            if algo == "inverse":
                buf = x.data.permute(0, 1, 3, 2)               # x is a permuted view of the [1,F,W,P] buffer
                assert buf.is_contiguous()                     # ... which the backward will read
                live = torch.nonzero(ok.permute(0, 1, 3, 2).reshape(-1))[7]
                buf.view(-1)[live] = float("nan")
            loss.backward()
            grads[algo] = [q.grad.clone() for q in lv]
        finally:
            ops.set_backward_algorithm("inverse")
    for a, b in zip(grads["inverse"], grads["checkpoint"]):
        assert torch.isfinite(a).all() and torch.equal(a, b)


def test_a_poisoned_graph_replay_does_not_stick(ta):
    """recorded into a HIP graph the walk-back call is replayed with the same token every time, so one
    poisoned replay (non-finite adjoint -> checkpoint fallback) would have sent every later replay to the fallback
    as well.  The forward's reduction kernel now clears the word.  Captured step = forward, a multiply of one live
    ray's saved x by a device scalar m, backward.  Replay with m = NaN -> the checkpoint kernel's gradients; replay
    with m = 1 -> the walk-back kernel's again (bit-equal to the eager walk-back step, and different bits from the
    checkpoint algorithm's)."""
    from torchoptics_amd import ops
    g = load_golden("G4_tessar_32x32")
    ins, mask, allow = dev_inputs(g)
    lv = [ins[i].clone().requires_grad_(True) for i in (5, 6, 7)]
    m = torch.ones((), device=DEV)

    def step():
        for q in lv:
            q.grad = None
        x, y, cx, cy, ok, back = ta.trace_skew(*ins[:5], *lv, mask, False, allow)
        loss = ta.compute_rms2d(x, y, ok)
        buf = x.data.permute(0, 1, 3, 2).view(-1)              # the [1,F,W,P] buffer the backward reads
        buf[live] = buf[live] * m
        loss.backward()
    with torch.no_grad():
        ok0 = ta.trace_skew(*ins[:5], *[q.detach() for q in lv], mask, False, allow)[4]
    live = torch.nonzero(ok0.permute(0, 1, 3, 2).reshape(-1))[7]
    ref = {}
    for algo in ("inverse", "checkpoint"):
        ops.set_backward_algorithm(algo)
        try:
            step()
            ref[algo] = [q.grad.clone() for q in lv]
        finally:
            ops.set_backward_algorithm("inverse")
    assert not all(torch.equal(a, b) for a, b in zip(ref["inverse"], ref["checkpoint"]))     # distinguishable
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(cap)
    for q in lv:
        q.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=cap):
        step()
    m.fill_(float("nan"))
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip([q.grad for q in lv], ref["checkpoint"]):
        assert torch.isfinite(a).all() and torch.equal(a, b)          # poisoned replay: the fallback did the work
    m.fill_(1.0)
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip([q.grad for q in lv], ref["inverse"]):
        assert torch.equal(a, b)                                      # clean replay: the walk-back kernel again


@pytest.mark.parametrize("wl", ["cfg3", "cfg3a", "cfg5"])
def test_full_size_properties(ta, wl):
    """BASELINE's configurations at full size (cfg3: 2^24 rays, 11 rows; cfg3a: the same with two aspheric rows;
    cfg5: 20 rows, 5 fields x 3 wavelengths, 15.7 M rays) through properties that need no oracle at that size:
    (a) tracing the two halves of the pupil separately gives the same per-ray outputs bit for bit and moments that
        add up to the full run's (this is also the multi-GPU sharding contract);
    (b) two runs are bitwise identical (fixed-order reductions, no atomics), forward and gradients;
    (c) the backward is linear in the upstream gradient of the moments: grad(a*g1 + b*g2) = a*grad(g1) + b*grad(g2)."""
    import bench
    args, meta, _ = bench.workload(wl, DEV, 1, 0, None)
    P = meta["P_local"]
    assert P * meta["F"] * meta["W"] >= 15 << 20
    extra = {k: args[k] for k in ("kappa", "poly") if k in args}

    def run(sl=slice(None), g_mom=None):
        lv = {k: args[k].detach().clone().requires_grad_(True) for k in ("c", "t", "mu")}
        x, y, cx, cy, ok, back = ta.trace_skew(args["x"][:, :, sl].contiguous(), args["y"][:, :, sl].contiguous(), args["z"],
                                               args["cx"], args["cy"], lv["c"], lv["t"], lv["mu"], args["mask"], **extra)
        mom = y._tl_spot[0]
        if g_mom is None:
            ta.compute_rms2d(x, y, ok).backward()
        else:
            mom.backward(g_mom)
        return (x, y, cx, cy, ok, back), mom.detach().clone(), [lv[k].grad.clone() for k in ("c", "t", "mu")]

    full, m_full, g_full = run()
    again, m_again, g_again = run()
    assert torch.equal(m_full, m_again) and all(torch.equal(a, b) for a, b in zip(g_full, g_again))          # (b)
    assert all(torch.equal(a, b) for a, b in zip(full, again))
    half = P // 2
    lo, m_lo, _ = run(slice(0, half))
    hi, m_hi, _ = run(slice(half, P))
    for f_, l_, h_ in zip(full, lo, hi):                                                                  # (a)
        assert torch.equal(f_[:, :, :half], l_) and torch.equal(f_[:, :, half:], h_)
    assert torch.allclose(m_lo + m_hi, m_full, rtol=1e-12, atol=1e-9)
    assert m_full[:, 3].sum().item() == full[4].sum().item()                  # sum of ok = moment 3, exactly
    gen = torch.Generator().manual_seed(5)
    g1 = torch.zeros_like(m_full)
    g2 = torch.zeros_like(m_full)
    g1[:, :3] = torch.randn(m_full.shape[0], 3, generator=gen, dtype=torch.float64).to(DEV) * 1e-6
    g2[:, :3] = torch.randn(m_full.shape[0], 3, generator=gen, dtype=torch.float64).to(DEV) * 1e-6
    _, _, ga = run(g_mom=g1)
    _, _, gb = run(g_mom=g2)
    _, _, gc_ = run(g_mom=0.7 * g1 - 1.9 * g2)
    for a, b, c_ in zip(ga, gb, gc_):                                                                     # (c)
        want = 0.7 * a.double() - 1.9 * b.double()
        scale = (0.7 * a.double().abs() + 1.9 * b.double().abs()).max().item()        # fp32 rounding of the terms
        assert (c_.double() - want).abs().max().item() <= 2e-5 * scale


@pytest.mark.parametrize("aggregate", [False, "sum"])
def test_backward_under_saved_tensor_hooks(ta, aggregate):
    """the backward reuses the tl_problem filled in forward; under saved-tensor hooks (save_on_cpu,
    activation checkpointing) the unpacked tensors live in OTHER storage, so the problem must be rebuilt from them.
    One step under torch.autograd.graph.save_on_cpu() gives the gradients of the plain step, bit for bit."""
    from torchoptics_amd import ray_tracing as rt
    g = load_golden("G4_tessar_32x32")
    res = []
    for hooks in (False, True):
        ins, mask, allow = dev_inputs(g)
        for q in ins[5:]:
            q.requires_grad_(True)                  # c, t, mu only: the walk-back backward
        ctxm = torch.autograd.graph.save_on_cpu() if hooks else torch.enable_grad()
        with ctxm:
            out = ta.trace_skew(*ins, mask, aggregate, allow)
            assert _used_walk_back(out[0])
            loss = ta.compute_rms2d(out[0], out[1], out[4])
            if aggregate:
                loss = loss + 0.2 * rt.penalty_sum(out[6], 8)
        loss.backward()
        res.append([q.grad.clone() for q in ins[5:]])
    for a, b in zip(*res):
        assert torch.equal(a, b)
