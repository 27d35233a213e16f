"""Double precision on the GPU: RayTracer(double_precision=True) (ray_tracing_lite.py:82-84 -- crashes in the reference,
SURVEY Appendix B3) and trace_skew on float64 tensors run the generic fp64 kernels (tl_trace_fwd_f64 / tl_trace_bwd_f64).
Checked against the oracle in fp64 (outputs to 1e-11 mm, masks equal, gradients to 1e-9) and against the reference's own
fp64 scalars and leaf gradients in the fixtures."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2
from test_oracle_asphere import asphere_params

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
IN = ("in_x", "in_y", "in_z", "in_cx", "in_cy", "in_c", "in_t", "in_mu")


@pytest.fixture(scope="module")
def ta():
    import torchoptics_amd
    from torchoptics_amd import _lib
    _lib.lib()
    return torchoptics_amd


@pytest.mark.parametrize("case", ["G2_cooke_16x16", "G4_tessar_32x32", "G5_cooke_failures", "G10_cooke_noback"])
@pytest.mark.parametrize("aspheric", [False, True])
def test_fp64_trace_matches_the_fp64_oracle(ta, case, aspheric):
    from oracle import trace_oracle as orc
    g = load_golden(case)
    allow = bool(g.get("allow_backward_rays", True))
    ins = [torch.from_numpy(g[n]).double() for n in IN]
    mask = torch.from_numpy(g["in_mask"])
    S = ins[5].shape[-1]
    kap0, pol0, kind = asphere_params(S, torch.float64)
    names = ["z", "cy", "c", "t", "mu"] + (["kappa", "poly"] if aspheric else [])
    gen = torch.Generator().manual_seed(11)
    wts = [torch.randn(g["x"].shape, generator=gen, dtype=torch.float64) * 1e-3 for _ in range(4)]

    def run(dev, tracer):
        lv = [ins[i].to(dev).clone().requires_grad_(True) for i in (2, 4, 5, 6, 7)]
        kw = {}
        if aspheric:
            lv += [kap0.to(dev).clone().requires_grad_(True), pol0.to(dev).clone().requires_grad_(True)]
        xin = ins[0].to(dev).clone().requires_grad_(True)
        if tracer == "oracle":
            o = orc.trace_skew_general(xin, ins[1].to(dev), lv[0], ins[3].to(dev), lv[1], lv[2], lv[3], lv[4], mask,
                                       *( (lv[5], lv[6], kind) if aspheric else ()), allow_backward_rays=allow)
            rms = orc.compute_rms2d(o[0], o[1], o[4])
        else:
            if aspheric:
                kw = dict(kappa=lv[5], poly=lv[6])
            o = ta.trace_skew(xin, ins[1].to(dev), lv[0], ins[3].to(dev), lv[1], lv[2], lv[3], lv[4], mask.to(dev), False, allow, **kw)
            assert o[0].dtype == torch.float64
            rms = ta.compute_rms2d(o[0], o[1], o[4])
        loss = rms + sum((q * w_.to(dev)).sum() for q, w_ in zip(o[:4], wts))      # moments seed + dense seeds
        loss.backward()
        return [q.detach().cpu() for q in o[:6]], [q.grad.cpu() for q in lv + [xin]], rms.item()
    got, gg, rms_g = run(DEV, "kernel")
    want, gw, rms_w = run("cpu", "oracle")
    assert torch.equal(got[4], want[4]) and bool((got[5] == want[5]).all())      # (the oracle's `back` may be un-broadcast over W)
    for i in range(4):
        assert (got[i] - want[i]).abs().max().item() < 1e-11, i
    assert abs(rms_g - rms_w) <= 1e-10 * abs(rms_w)          # sums of ~1e4 rays in different orders
    for n, a, b in zip(names + ["x_in"], gg, gw):
        assert rel_l2(a.numpy(), b.numpy()) < 1e-9, f"{case} asph={aspheric} d/d{n}: {rel_l2(a.numpy(), b.numpy()):.2e}"


def test_raytracer_double_precision_reproduces_the_reference_fp64_numbers(ta):
    """RayTracer(double_precision=True) end to end on fp32 leaves: rms and d rms / d(c, t, nd, v) against the reference's
    own fp64 evaluation (fixture G2: rms64, g_c64 ...)."""
    import yaml_free_lenses as L
    g = load_golden("G2_cooke_16x16")
    lens, specs, leaves = L.build("cooke", DEV)
    tr = ta.RayTracer(mode="circular", n_rays=(16, 16), rel_fields=(0., 0.707, 1.), wavelengths=("C", "d", "F"),
                      double_precision=True, default_device=DEV)
    x, y, cx, cy, ok, back = tr.trace_rays(specs, lens)
    assert x.dtype == torch.float64
    rms = ta.compute_rms2d(x, y, ok)
    assert rms.dtype == torch.float64
    # the fixture's fp64 run starts from fp64 leaves; here the fp32 leaves are widened, the fan is the fp32 grid widened
    assert abs(rms.item() - float(g["rms64"])) <= 2e-6 * float(g["rms64"])
    rms.backward()
    for k in ("c", "t", "nd"):
        e = rel_l2(leaves[k].grad.cpu().numpy(), g["g_" + k + "64"])
        assert e < 2e-5, f"d/d{k}: {e:.2e}"          # dominated by the fp32 -> fp64 widening of the inputs (1e-7 relative in nd, c)


def test_double_precision_with_ray_aiming_and_a_lens_batch(ta):
    """double_precision with one ray-aiming iteration (the op sequence, through the fp64 trace with per-ray input gradients)
    and a padded batch of two lenses: finite, close to the fp32 path, per-lens gradients present."""
    import yaml_free_lenses as L
    from test_gpu_batch import _batch
    res = {}
    for dp in (False, True):
        lens, specs, leaves = L.build("cooke", DEV)
        tr = ta.RayTracer(mode="circular", n_rays=(16, 16), rel_fields=(0., 0.707, 1.), wavelengths=("C", "d", "F"),
                          n_ray_aiming_iter=1, double_precision=dp, default_device=DEV)
        x, y, cx, cy, ok, back = tr.trace_rays(specs, lens)
        rms = ta.compute_rms2d(x, y, ok)
        rms.backward()
        res[dp] = (rms.item(), leaves["c"].grad.clone())
    assert abs(res[True][0] - res[False][0]) < 2e-5 * res[False][0]
    assert rel_l2(res[True][1].cpu().numpy(), res[False][1].cpu().numpy()) < 1e-3
    lens, specs, leaves = _batch(DEV)
    tr = ta.RayTracer(mode="circular", n_rays=(16, 16), rel_fields=(0., 0.707, 1.), wavelengths=("C", "d", "F"),
                      double_precision=True, default_device=DEV)
    out = tr.trace_rays(specs, lens)
    from torchoptics_amd import ray_tracing as rt
    per_lens = rt.compute_rms2d_batch(out[0], out[1], out[4])
    assert per_lens.shape == (2,) and per_lens.dtype == torch.float64
    per_lens.sum().backward()
    assert torch.isfinite(leaves["c"].grad).all() and leaves["c"].grad[:7].abs().max() > 0 and leaves["c"].grad[7:].abs().max() > 0


def test_double_precision_refuses_what_it_does_not_have(ta):
    g = load_golden("G2_cooke_16x16")
    ins = [torch.from_numpy(g[n]).double().to(DEV) for n in IN]
    mask = torch.from_numpy(g["in_mask"]).to(DEV)
    with pytest.raises(NotImplementedError):
        ta.trace_skew(*ins, mask, True, True)
