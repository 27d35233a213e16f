"""B > 1: a padded batch of two lenses of different length (the reference's containers exist for this; its
`trace_skew` broadcasts over dim 0).  One launch per lens; outputs equal the single-lens traces bit for bit, the
strict forward equals the IEEE oracle on the padded batch, compute_rms2d reads lens 0 only, gradients flow to both."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _batch(device, dtype=torch.float32):
    import yaml_free_lenses as L
    from torchoptics_amd import lens_modeling as lm
    a, b = L.PRESCRIPTIONS["cooke"], L.PRESCRIPTIONS["doublet"]
    st = lm.Structure(stop_idx=np.array(a["stop_idx"] + b["stop_idx"]), sequence=np.array(a["sequence"] + b["sequence"]),
                      default_device=device)
    leaves = {k: torch.tensor(a[k] + b[k], dtype=dtype, device=device, requires_grad=True) for k in ("c", "t", "nd", "v")}
    lens = lm.Lens(st, leaves["c"], leaves["t"], leaves["nd"], leaves["v"])
    specs = lm.Specs(st, torch.tensor([L.EPD, L.EPD], dtype=dtype, device=device),
                     torch.tensor([np.deg2rad(L.HFOV_DEG)] * 2, dtype=dtype, device=device))
    return lens, specs, leaves


def test_two_lens_padded_batch():
    import yaml_free_lenses as L
    import torchoptics_amd as ta
    from oracle import trace_oracle as orc
    lens, specs, leaves = _batch(DEV)
    assert lens.c.shape == (2, 7) and not lens.structure.mask[1, 5:].any()
    kw = dict(mode="circular", n_rays=(16, 16), rel_fields=(0., 0.707, 1.), wavelengths=("C", "d", "F"))
    tr = ta.RayTracer(default_device=DEV, **kw)
    a = tr.assemble(specs, lens)
    out = ta.trace_skew(a["x"], a["y"], a["z"], a["cx"], a["cy"], a["c"], a["t"], a["mu"], a["mask"])
    assert out[0].shape == (2, 3, 256, 3) and out[4].dtype == torch.bool
    # (1) the strict forward on the padded batch = the IEEE oracle on the same padded batch
    cpu = {k: v.detach().cpu() for k, v in a.items()}
    want = orc.trace_skew(cpu["x"], cpu["y"], cpu["z"], cpu["cx"], cpu["cy"], cpu["c"], cpu["t"], cpu["mu"], cpu["mask"],
                          ieee_sqrt=True)
    for i in range(6):
        assert torch.equal(out[i].cpu(), want[i]), i
    # (2) each lens of the batch = that lens traced alone (up to the pupil position z: one lens on the GPU takes
    #     the fp64 tl_pupil_position kernel, a batch the fp32 ABCD chain, so z differs by a rounding)
    for b, name in enumerate(("cooke", "doublet")):
        l1, s1, _ = L.build(name, DEV)
        o1 = tr.trace_rays(s1, l1)
        for i in range(4):
            assert (out[i][b:b + 1] - o1[i]).abs().max().item() < 5e-6, (name, i)
        assert torch.equal(out[4][b:b + 1], o1[4]) and torch.equal(out[5][b:b + 1], o1[5])
    # (3) compute_rms2d reads sample 0 (ray_tracing_lite.py:695,699) and back-propagates to lens 0's rows only;
    #     a loss on lens 1's rays reaches lens 1's rows
    l0, s0, lv0 = L.build("cooke", DEV)
    o0 = tr.trace_rays(s0, l0)
    rms0 = ta.compute_rms2d(o0[0], o0[1], o0[4])
    rms = ta.compute_rms2d(out[0], out[1], out[4])
    assert abs(rms.item() - rms0.item()) <= 1e-5 * rms0.item()
    (rms + out[1][1].square().mean()).backward()
    rms0.backward()
    g = leaves["c"].grad
    assert torch.allclose(g[:7], lv0["c"].grad, rtol=2e-3, atol=1e-6) and g[7:].abs().max().item() > 0


def test_aggregate_on_a_batch_is_refused():
    import torchoptics_amd as ta
    lens, specs, _ = _batch(DEV)
    tr = ta.RayTracer(mode="circular", n_rays=(8, 8), rel_fields=(0., 1.), wavelengths=("d",), default_device=DEV)
    with pytest.raises(NotImplementedError):
        tr.trace_rays(specs, lens, aggregate=True)
