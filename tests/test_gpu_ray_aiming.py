"""tl_ray_aim (one kernel: marginal ray, tee rays + Jacobian, Newton step, affine pupil map) against the reference's
sequence of tensor ops (RayTracer.ray_aiming with the kernel switched off: two traces through the HIP tracer + autograd,
ray_tracing_lite.py:129-208) and against the reference's own aimed coordinates (fixture G6)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ta():
    import torchoptics_amd
    from torchoptics_amd import _lib
    _lib.lib()
    return torchoptics_amd


def _aimed(tr, specs, lens, kernel):
    from torchoptics_amd import ray_tracing as rt
    rt.set_ray_aiming_kernel(kernel)
    try:
        with torch.no_grad():
            a = tr.assemble(specs, lens)
    finally:
        rt.set_ray_aiming_kernel(True)
    return a["x"], a["y"]


@pytest.mark.parametrize("name,kw", [("cooke", {}), ("tessar", {}), ("doublet", {}), ("cooke", dict(epd=12.0, hfov_deg=32.0))])
def test_aiming_kernel_matches_the_reference_sequence_of_ops(ta, name, kw):
    import yaml_free_lenses as L
    lens, specs, _ = L.build(name, DEV, grad=False, **kw)
    tr = ta.RayTracer(mode="circular", n_rays=(16, 16), rel_fields=(0., 0.5, 0.707, 1.), wavelengths=("C", "d", "F"),
                      n_ray_aiming_iter=1, default_device=DEV)
    xk, yk = _aimed(tr, specs, lens, True)
    xr, yr = _aimed(tr, specs, lens, False)
    assert xk.shape == xr.shape == (1, 4, 256, 3)
    assert (xk - xr).abs().max().item() < 2e-5 * xr.abs().max().item()
    assert (yk - yr).abs().max().item() < 2e-5 * yr.abs().max().item()
    # ... and the aiming does something: the aimed fan differs from the plain one
    tr0 = ta.RayTracer(mode="circular", n_rays=(16, 16), rel_fields=(0., 0.5, 0.707, 1.), wavelengths=("C", "d", "F"),
                       n_ray_aiming_iter=0, default_device=DEV)
    with torch.no_grad():
        a0 = tr0.assemble(specs, lens)
    assert (yk - a0["y"]).abs().max().item() > 1e-3


def test_aiming_kernel_reproduces_the_reference_fixture(ta):
    """G6: the reference's own aimed pupil coordinates of the Cooke triplet (16 x 16, 3 fields, C d F)."""
    import yaml_free_lenses as L
    g = load_golden("G6_cooke_aim1")
    lens, specs, _ = L.build("cooke", DEV, grad=False)
    tr = ta.RayTracer(mode="circular", n_rays=(16, 16), rel_fields=(0., 0.707, 1.), wavelengths=("C", "d", "F"),
                      n_ray_aiming_iter=1, default_device=DEV)
    x, y = _aimed(tr, specs, lens, True)
    assert np.abs(x.cpu().numpy() - g["in_x"]).max() < 2e-5 and np.abs(y.cpu().numpy() - g["in_y"]).max() < 2e-5


def test_aiming_kernel_on_a_lens_batch_with_aspheres_and_dead_rays(ta):
    """B = 3 lenses in one launch: the two-asphere double Gauss (aspheric row 1 in front of the stop), its all-spherical
    twin, and one stopped so wide that tee rays die on the way (no step for those, as in the reference)."""
    from torchoptics_amd import lens_modeling as lm, prescriptions as P
    la, sa, _ = P.double_gauss(DEV, requires_grad=False, aspheres=True)
    ls, ss, _ = P.double_gauss(DEV, requires_grad=False)
    for lens, specs in ((la, sa), (ls, ss), (ls, lm.Specs(ls.structure, ss.epd * 3.2, ss.hfov))):
        tr = ta.RayTracer(mode="circular", n_rays=(8, 8), rel_fields=(0., 1.), wavelengths=("d", "F"), n_ray_aiming_iter=1,
                          default_device=DEV)
        xk, yk = _aimed(tr, specs, lens, True)
        xr, yr = _aimed(tr, specs, lens, False)
        assert torch.isfinite(xk).all() and torch.isfinite(yk).all()
        assert (xk - xr).abs().max().item() < 3e-5 * xr.abs().max().item()
        assert (yk - yr).abs().max().item() < 3e-5 * yr.abs().max().item()


def test_minibatch_of_lenses_through_the_aiming_kernel(ta):
    """256 perturbed Cooke triplets (the real caller's minibatch): per-lens losses with the kernel = with the op sequence."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import minibatch_loss as mb
    from torchoptics_amd import ray_tracing as rt
    st, specs, leaves, n_seq = mb.build_batch(64, DEV)
    tr = ta.RayTracer(mode="circular", n_rays=(8, 8), rel_fields=mb.FIELDS, wavelengths=mb.WAVELENGTHS, n_ray_aiming_iter=1,
                      default_device=DEV)
    res = []
    for kernel in (True, False):
        rt.set_ray_aiming_kernel(kernel)
        try:
            leaves["c"].grad = leaves["t"].grad = None
            lens = ta.Lens(st, leaves["c"], leaves["t"], leaves["nd"], leaves["v"])
            ld = rt.unsupervised_loss_batch(tr.trace_rays(specs, lens, aggregate="sum"), n_seq, 0.2)
            ld["loss_unsup"].sum().backward()
            res.append((ld["loss_unsup"].detach().clone(), leaves["c"].grad.clone()))
        finally:
            rt.set_ray_aiming_kernel(True)
    (l1, g1), (l2, g2) = res
    assert ((l1 - l2).abs() / l2.abs()).max().item() < 2e-5
    assert ((g1 - g2).norm() / g2.norm()).item() < 1e-3          # the penalty gradient's own fp32 noise level


@pytest.mark.parametrize("name", ["cooke", "tessar"])
def test_one_launch_fan_equals_remap_clamp_scale_bit_for_bit(ta, name):
    """tl_aim_fan (the aimed, clamped, pupil-scaled fan in one launch, laid out [B,F,W,P]) against the three tensor-op steps of
    RayTracer.assemble it replaces -- remap, torch.clamp(-2, 2), scale_to_epd -- on a fan wide enough for the clamp to act."""
    import yaml_free_lenses as L
    from torchoptics_amd import ray_tracing as rt
    lens, specs, _ = L.build(name, DEV, grad=False, epd=12.0, hfov_deg=30.0)
    tr = ta.RayTracer(mode="circular", n_rays=(16, 16), rel_fields=(0., 0.5, 0.707, 1.), wavelengths=("C", "d", "F"),
                      n_ray_aiming_iter=1, default_device=DEV)
    aim = tr.ray_aiming(specs, lens.detach(), True)
    assert callable(getattr(aim, "fan", None))
    xp, yp = rt.circle(None, 16, 16, DEV)
    xp, yp = 2.5 * xp, 2.5 * yp                                # beyond the clamp for part of the grid
    fx, fy = aim.fan(xp, yp, specs.epd)
    ox, oy = (rt.scale_to_epd(torch.clamp(v, -2, 2), specs.epd) for v in aim(xp, yp))
    assert fx.shape == ox.shape and fx.stride(2) == 1           # consecutive pupil points are contiguous
    assert torch.equal(fx, ox) and torch.equal(fy, oy)
    assert (torch.clamp(aim(xp, yp)[0], -2, 2) != aim(xp, yp)[0]).any()
    # not a shared [1,1,P,1] fan: the caller composes it from tensor ops
    assert aim.fan(xp.expand(1, 4, -1, 1), yp.expand(1, 4, -1, 1), specs.epd) is None
    # and the assembled fan of trace_rays is that one-launch fan
    a = tr.assemble(specs, lens)
    xq, yq = rt.circle(None, 16, 16, DEV)
    gx, gy = aim.fan(xq, yq, specs.epd)
    assert torch.equal(a["x"], gx) and torch.equal(a["y"], gy) and a["x"].stride(2) == 1
