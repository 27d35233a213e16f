"""Soft-histogram PSF (torchoptics_amd.metrics.compute_psf).  The reference holds it only as TensorFlow text
(ray_tracing.py:206-270), which cannot run here: PARITY UNPINNED.  What is checked: the GEMM formulation against a
loop evaluation (`_by_the_text`) that follows the reference text statement by statement -- including its grid extent,
which subtracts y_target a second time from the already centred y (ray_tracing.py:231) -- on fans with off-axis
fields (y_target up to 3 mm), the corrected extent as an option, and the properties the definition implies."""
import numpy as np
import torch

from torchoptics_amd import metrics


def _by_the_text(x, y, n_bins, increment, y_extent="reference"):
    """ray_tracing.py:206-270 statement by statement, with loops (fp64 numpy): y_target = mean, y centred on it, then
    (increment None) y_min / y_max of the CENTRED y and y_size = 2 max(y_max - y_target, y_target - y_min) -- y_target
    subtracted a second time, as the text has it; grid, half-x evaluation, mirror, normalise.
    Returns (kernels, y_size)."""
    L, F, W, R = x.shape
    nxb, nyb = n_bins
    out = np.zeros((L * F, W, nyb, nxb))
    sizes = np.zeros(L * F)
    for g in range(L * F):
        xg, yg = x.reshape(L * F, W, R)[g], y.reshape(L * F, W, R)[g]
        yt = yg.mean()                                   # :218
        yc = yg - yt                                     # :221
        if increment is None:
            y_min, y_max = yc.min(), yc.max()            # :229-230 (of the centred y)
            xs = xg.max()                                # :231
            ys = 2 * max(y_max - yt, yt - y_min) if y_extent == "reference" else 2 * max(y_max, -y_min)      # :232
            xi, yi = xs / nxb, ys / nyb
        else:
            xi = yi = increment
            ys = increment * nxb                         # :226 (sic: n_x_bins)
        sizes[g] = ys
        cx = (np.arange(nxb // 2 + 1) if nxb % 2 else np.arange(nxb // 2) + 0.5) * xi
        cy = (np.arange(nyb) + 0.5 - nyb / 2) * yi
        for w in range(W):
            k = np.zeros((nyb, len(cx)))
            for iy in range(nyb):
                for ix in range(len(cx)):
                    k[iy, ix] = np.sum(np.exp(-((xg[w] - cx[ix]) / (xi / 2)) ** 2 / 2) * np.exp(-((yc[w] - cy[iy]) / (yi / 2)) ** 2 / 2))
            full = np.concatenate((k[:, :0:-1], k), axis=1) if nxb % 2 else np.concatenate((k[:, ::-1], k), axis=1)
            out[g, w] = full / full.sum()
    return out, sizes


def _fan(seed=0, F=2, W=3, R=400):
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 0.01, (1, F, W, R))
    x = np.concatenate((x, -x), axis=-1)                       # symmetric in x, as the half-pupil samplers give
    y = rng.normal(0, 0.02, (1, F, W, R)) + np.linspace(0, 3, F)[None, :, None, None]
    y = np.concatenate((y, y), axis=-1)
    return x, y


def test_psf_follows_the_reference_text_including_its_grid_extent():
    x, y = _fan()                                      # two fields: y_target 0 and 3 mm
    for n_bins, inc in (((21, 21), None), ((8, 10), None), ((15, 15), 0.004)):
        xs, ys, yt, k, acc = metrics.compute_psf(torch.from_numpy(x), torch.from_numpy(y), n_bins=n_bins, increment=inc)
        want, want_ys = _by_the_text(x, y, n_bins, inc)
        assert k.shape == want.shape
        assert np.abs(k.numpy() - want).max() < 1e-12
        assert np.allclose(yt.numpy(), y.reshape(2, -1).mean(axis=1))
        assert np.allclose(np.broadcast_to(np.asarray(ys), (2,)), want_ys, rtol=1e-13)
        if inc is None:
            # the off-axis field's grid spans ~2 x 3 mm although its spot is +-0.1 mm: the reference's double subtraction
            assert 5.9 < float(ys[1]) < 6.3 and float(ys[0]) < 0.3


def test_psf_centred_extent_is_an_explicit_deviation():
    x, y = _fan()
    xs, ys, yt, k, acc = metrics.compute_psf(torch.from_numpy(x), torch.from_numpy(y), n_bins=(21, 21), y_extent="centred")
    want, want_ys = _by_the_text(x, y, (21, 21), None, y_extent="centred")
    assert np.abs(k.numpy() - want).max() < 1e-12 and np.allclose(ys.numpy(), want_ys, rtol=1e-13)
    assert float(ys[1]) < 0.3                          # both grids now sized on their own spot


def test_psf_properties():
    x, y = _fan(seed=1)
    xs, ys, yt, k, acc = metrics.compute_psf(torch.from_numpy(x), torch.from_numpy(y), n_bins=(21, 21), y_extent="centred")
    assert torch.allclose(k.sum(dim=(-1, -2)), torch.ones(2, 3, dtype=k.dtype))           # unit area per channel
    assert torch.allclose(k, torch.flip(k, dims=(-1,)))                                    # mirrored in x
    # centroid of the histogram in y = centroid of the rays (the grid is centred on it), within a pixel fraction
    cy = (torch.arange(21, dtype=k.dtype) + 0.5 - 10.5)[None, None, :] * (ys / 21)[:, None, None]
    cen = (k.sum(dim=-1) * cy).sum(dim=-1)
    assert (cen.abs() < 0.25 * (ys / 21)[:, None]).all()
    assert ((acc > 0.8) & (acc <= 1)).all()          # the grid spans the y extent; in x it ends at the outermost ray
    # weights: dropping half of the rays by weight = evaluating on that half
    w = torch.zeros(1, 2, 3, x.shape[-1], dtype=torch.float64)
    w[..., ::2] = 1
    a = metrics.compute_psf(torch.from_numpy(x), torch.from_numpy(y), n_bins=(21, 21), increment=0.004,
                            y_target=torch.from_numpy(y.reshape(2, -1).mean(axis=1)), weights=w)[3]
    b = metrics.compute_psf(torch.from_numpy(x[..., ::2]), torch.from_numpy(y[..., ::2]), n_bins=(21, 21), increment=0.004,
                            y_target=torch.from_numpy(y.reshape(2, -1).mean(axis=1)))[3]
    assert torch.allclose(a, b, atol=1e-12)


def test_psf_from_trace_layout():
    x, y = _fan(seed=2)
    xt, yt = torch.from_numpy(x).permute(0, 1, 3, 2), torch.from_numpy(y).permute(0, 1, 3, 2)     # [1,F,P,W] like the tracer
    ok = torch.ones_like(xt, dtype=torch.bool)
    a = metrics.psf_from_trace(xt, yt, ok)[3]
    b = metrics.compute_psf(torch.from_numpy(x), torch.from_numpy(y), y_extent="centred")[3]
    assert torch.allclose(a, b, atol=1e-12)
