"""Pupil samplers that the reference only has as unreachable TensorFlow text (parity unpinned):
property tests of the patterns, and the RayTracer mode table."""
import numpy as np
import pytest
import torch

import torchoptics_amd as ta
from torchoptics_amd import ray_tracing as rt


def test_simple_line_samplers():
    x, y = rt.chief(None, None, "cpu")
    assert x.shape == (1, 1, 1, 1) and x.item() == 0 and y.item() == 0
    x, y = rt.meridional_uniform(None, 5, "cpu")
    assert torch.equal(y.flatten(), torch.tensor([-1., -.5, 0., .5, 1.])) and not x.any()
    x, y = rt.sagittal_uniform(None, 3, "cpu")
    assert torch.equal(x.flatten(), torch.tensor([0., .5, 1.])) and not y.any()
    x, y = rt.circle_outer_edge_uniform(None, 8, "cpu")
    assert torch.allclose(x * x + y * y, torch.ones_like(x), atol=1e-6) and x.flatten()[0] == 1


@pytest.mark.parametrize("fn", [rt.skew_uniform_half_equidistant, rt.skew_uniform_half_jittered])
def test_half_pupil_shell_patterns(fn):
    n_r, n_i = 4, 3
    x, y = fn(None, n_r, n_i, "cpu")
    assert x.shape == (1, 1, n_i * n_r ** 2, 1)
    r = torch.sqrt(x * x + y * y).flatten()
    assert (x >= -1e-7).all() and r.max() <= 1 + 1e-6                 # right half of the unit pupil
    radii = np.unique(np.round(r.numpy(), 5))
    if fn is rt.skew_uniform_half_equidistant:
        assert np.allclose(radii, (np.arange(n_r) + 0.5) / n_r, atol=1e-5)
        counts = [int(((r - rv).abs() < 1e-5).sum()) for rv in radii]
        assert counts == [n_i * (2 * i + 1) for i in range(n_r)]      # equal area per ray
    else:
        assert np.isclose(radii.max(), 1.0, atol=1e-5) and np.isclose(radii.min(), 0.0, atol=1e-5)
        assert len(radii) == 2 * n_r


def test_inner_square_and_vignetting():
    x, y = rt.skew_inner_square_half(None, 4, None, "cpu")
    assert x.shape == (1, 1, 16, 1) and (x > 0).all() and (x * x + y * y <= 1 + 1e-6).all()
    assert abs(y.max().item() - 1 / np.sqrt(2)) < 1e-6
    yy = torch.tensor([-1., 0., 1.]).reshape(1, 1, 3, 1)
    up, down = torch.tensor([[0.2]]), torch.tensor([[0.0]])           # upper edge in by 0.2
    out = rt.apply_vignetting(yy, up, down).flatten()
    assert torch.allclose(out, torch.tensor([-1., -0.1, 0.8]))


def test_every_mode_of_the_reference_table_is_constructible():
    for mode, n in (("skew_random", (4, 4)), ("circular", (4, 4)), ("tee", (1, 1)), ("chief", 1),
                    ("meridional_uniform", 7), ("sagittal_uniform", 7), ("skew_outer_edge_uniform", 12),
                    ("skew_uniform_half_equidistant", (3, 2)), ("skew_uniform_half_jittered", (3, 2)),
                    ("skew_inner_square_half", (4, None))):
        tr = ta.RayTracer(mode=mode, n_rays=n, default_device="cpu")
        x, y = tr.pupil_span(torch.zeros(1, 1, 1, 1))
        assert x.shape == y.shape and x.dim() == 4 and x.shape[:2] == (1, 1)
    with pytest.raises(ValueError):
        ta.RayTracer(mode="nonsense")
