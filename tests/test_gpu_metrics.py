"""Post-trace metrics (reference: unreachable TF text, parity unpinned) on the GPU: physical sanity."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def setup():
    import torchoptics_amd as ta
    from torchoptics_amd import _lib, prescriptions as P
    _lib.lib()
    lens, specs, _ = P.double_gauss(DEV, requires_grad=False)
    return ta, lens, specs


def test_distortion_of_a_symmetric_double_gauss_is_small(setup):
    ta, lens, specs = setup
    d = ta.metrics.compute_distortion(specs, lens, [0.3, 0.7, 1.0], DEV)
    assert d.shape == (1, 3) and torch.isfinite(d).all()
    assert d.abs().max().item() < 0.05                       # a (nearly) symmetric form: well under 5 %
    assert d.abs()[0, 0] <= d.abs()[0, 2] + 1e-4             # grows with field


def test_relative_illumination_starts_at_one_and_falls_off(setup):
    ta, lens, specs = setup
    ri = ta.metrics.compute_relative_illumination(specs, lens, [0., 0.5, 1.0], None, 1, ('C', 'd', 'F'), DEV)
    assert ri.shape == (1, 3, 3)
    # normalised to the on-axis value of the FIRST wavelength (as the reference's formula does); the other
    # colours differ by the chromatic variation of the image-space aperture
    assert abs(ri[0, 0, 0].item() - 1.0) < 1e-5 and (ri[:, 0, :] - 1).abs().max().item() < 0.02
    assert (ri[:, 2, :] < ri[:, 1, :]).all() and (ri[:, 1, :] < 1.0).all() and (ri[:, 2, :] > 0.5).all()
    cos4 = np.cos(np.deg2rad(14.0)) ** 4                      # natural fall-off at the edge of a 14 deg field
    assert abs(ri[0, 2, 1].item() - cos4) < 0.08


def test_ray_aiming_reduces_the_aiming_error(setup):
    ta, lens, specs = setup
    fields = [0., 0.7, 1.0]
    e0 = ta.metrics.compute_ray_aiming_error(specs, lens, fields, None, 0, 'real', DEV)
    e1 = ta.metrics.compute_ray_aiming_error(specs, lens, fields, None, 1, 'real', DEV)
    assert e0.shape == e1.shape == (1, 3, 2, 1)
    assert e1.abs().max().item() < 0.25 * e0.abs().max().item() + 1e-6
    assert e0[:, 0].abs().max().item() < 1e-5                # on axis the unaimed marginal rays already hit the rim


def test_vignetting_function_shrinks_the_traced_pupil(setup):
    ta, lens, specs = setup
    import dataclasses
    sp = dataclasses.replace(specs, vig_up=torch.tensor([0.3], device=DEV), vig_down=torch.tensor([0.1], device=DEV),
                             vig_x=torch.tensor([0.0], device=DEV))
    lin = lambda fields, v: fields * v[:, None]               # noqa: E731  vignetting grows linearly with the field
    tr = ta.RayTracer(mode='meridional_uniform', n_rays=21, rel_fields=(0., 1.), wavelengths=('d',), vig_fn=lin,
                      default_device=DEV)
    a = tr.assemble(sp, lens)
    y = a['y'][0] / (specs.epd / 2)                           # [F, P, 1] relative pupil height
    assert torch.allclose(y[0].max(), torch.tensor(1.0, device=DEV)) and torch.allclose(y[0].min(), torch.tensor(-1.0, device=DEV))
    assert abs(y[1].max().item() - 0.7) < 1e-6 and abs(y[1].min().item() + 0.9) < 1e-6
    out = tr.trace_rays(sp, lens)
    assert out[4].all().item()


def test_psf_of_a_traced_fan(setup):
    """Soft-histogram PSF (parity unpinned: TF text only) on the kernel's own outputs: unit area per channel, its
    grid is centred on the y centroid the fused spot moments give, failed rays are left out through ray_ok."""
    ta, lens, specs = setup
    tr = ta.RayTracer(mode="circular", n_rays=(32, 64), rel_fields=(0., 0.7, 1.0), wavelengths=("C", "d", "F"),
                      default_device=DEV)
    x, y, cx, cy, ok, back = tr.trace_rays(specs, lens)
    xs, ys, yt, k, acc = ta.metrics.psf_from_trace(x, y, ok, n_bins=(21, 21))
    assert k.shape == (3, 3, 21, 21) and torch.isfinite(k).all()
    assert torch.allclose(k.sum(dim=(-1, -2)), torch.ones(3, 3, device=DEV), atol=1e-5)
    mom = y._tl_spot[0]                                     # [F, 10] fused moments of the same trace
    assert torch.allclose(yt.double(), mom[:, 1] / mom[:, 3], rtol=1e-6, atol=1e-7)     # sum ok*y / sum ok
    assert (acc > 0.8).all()
    # a fixed 2 um pixel: the on-axis spot of this f/3 lens sits inside 21 pixels
    k2 = ta.metrics.psf_from_trace(x, y, ok, n_bins=(21, 21), increment=0.002)[3]
    assert k2[0, 1, 8:13, 8:13].sum().item() > 0.5
