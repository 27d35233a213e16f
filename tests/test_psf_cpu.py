"""Soft-histogram PSF (torchoptics_amd.metrics.compute_psf).  The reference holds it only as TensorFlow text
(ray_tracing.py:206-270), which cannot run here: PARITY UNPINNED.  What is checked: the GEMM formulation against a
literal loop evaluation of the reference's formula, and the properties the definition implies."""
import numpy as np
import torch

from torchoptics_amd import metrics


def _literal(x, y, n_bins, increment):
    """The reference's arithmetic written out with loops (fp64 numpy): grid, half-x evaluation, mirror, normalise."""
    L, F, W, R = x.shape
    nxb, nyb = n_bins
    out = np.zeros((L * F, W, nyb, nxb))
    for g in range(L * F):
        xg, yg = x.reshape(L * F, W, R)[g], y.reshape(L * F, W, R)[g]
        yt = yg.mean()
        yc = yg - yt
        if increment is None:
            xs, ys = xg.max(), 2 * max(yc.max(), -yc.min())
            xi, yi = xs / nxb, ys / nyb
        else:
            xi = yi = increment
        cx = (np.arange(nxb // 2 + 1) if nxb % 2 else np.arange(nxb // 2) + 0.5) * xi
        cy = (np.arange(nyb) + 0.5 - nyb / 2) * yi
        for w in range(W):
            k = np.zeros((nyb, len(cx)))
            for iy in range(nyb):
                for ix in range(len(cx)):
                    k[iy, ix] = np.sum(np.exp(-((xg[w] - cx[ix]) / (xi / 2)) ** 2 / 2) * np.exp(-((yc[w] - cy[iy]) / (yi / 2)) ** 2 / 2))
            full = np.concatenate((k[:, :0:-1], k), axis=1) if nxb % 2 else np.concatenate((k[:, ::-1], k), axis=1)
            out[g, w] = full / full.sum()
    return out


def _fan(seed=0, F=2, W=3, R=400):
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 0.01, (1, F, W, R))
    x = np.concatenate((x, -x), axis=-1)                       # symmetric in x, as the half-pupil samplers give
    y = rng.normal(0, 0.02, (1, F, W, R)) + np.linspace(0, 3, F)[None, :, None, None]
    y = np.concatenate((y, y), axis=-1)
    return x, y


def test_psf_matches_the_literal_formula():
    x, y = _fan()
    for n_bins, inc in (((21, 21), None), ((8, 10), None), ((15, 15), 0.004)):
        xs, ys, yt, k, acc = metrics.compute_psf(torch.from_numpy(x), torch.from_numpy(y), n_bins=n_bins, increment=inc)
        want = _literal(x, y, n_bins, inc)
        assert k.shape == want.shape
        assert np.abs(k.numpy() - want).max() < 1e-12
        assert np.allclose(yt.numpy(), y.reshape(2, -1).mean(axis=1))


def test_psf_properties():
    x, y = _fan(seed=1)
    xs, ys, yt, k, acc = metrics.compute_psf(torch.from_numpy(x), torch.from_numpy(y), n_bins=(21, 21))
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
    b = metrics.compute_psf(torch.from_numpy(x), torch.from_numpy(y))[3]
    assert torch.allclose(a, b, atol=1e-12)
