"""
Image-quality metrics computed from a handful of traced rays: the step AFTER the hot path.

In the reference these exist only as commented-out TensorFlow code (ray_tracing_lite.py:848-934;
originals in ray_tracing.py:815-901) and are unreachable in its PyTorch port.  Written here from their
stated intent on top of RayTracer.trace_rays (i.e. the HIP kernels); PARITY UNPINNED by the reference.
"""
from __future__ import annotations

import torch

from . import paraxial
from .ray_tracing import RayTracer, apply_vignetting


def compute_distortion(specs, lens, relative_fields, default_device="cuda"):
    """Relative distortion (y_chief - y_ref) / y_ref per field, d line.

    y_ref = EFL tan(field) corrected for the defocus of the image plane from the paraxial focus
    (last gap - BFL) along the traced chief-ray direction.  Field 0 gives 0/0 = NaN; pass fields > 0.
    """
    tr = RayTracer(mode='chief', n_rays=1, rel_fields=relative_fields, wavelengths=['d'], vig_fn=None,
                   default_device=default_device)
    _, y, _, cy, *_ = tr.trace_rays(specs, lens)
    n_lens = len(specs)
    y, cy = y.reshape(n_lens, -1), cy.reshape(n_lens, -1)
    fields = torch.tensor(list(relative_fields), dtype=y.dtype, device=y.device)
    efl, bfl = paraxial.get_first_order(lens)
    ideal = torch.tan(fields[None, :] * specs.hfov[:, None]) * efl[:, None]
    rows = torch.arange(n_lens, device=y.device)
    last = lens.structure.mask_torch.sum(dim=1) - 1
    defocus = lens.t[rows, last] - bfl
    ref_y = ideal + defocus[:, None] * cy / torch.sqrt(1 - cy * cy)
    return (y - ref_y) / ref_y


def compute_relative_illumination(specs, lens, relative_fields, vig_fn=None, n_ray_aiming_iter=1,
                                  wavelengths=('d',), default_device="cuda"):
    """Relative illumination per field and wavelength from the image-space solid angle of three rays
    (upper / lower marginal and one sagittal ray; Rimmer, doi 10.1117/12.938414), normalised to the
    on-axis field (relative_fields[0] must be 0).  Fields with a failed ray report 1."""
    assert relative_fields[0] == 0., "the first field must be the axis"
    tr = RayTracer(mode='tee', rel_fields=relative_fields, vig_fn=vig_fn, n_ray_aiming_iter=n_ray_aiming_iter,
                   wavelengths=wavelengths, default_device=default_device)
    x = torch.tensor([0., 0., 1.], device=default_device).reshape(1, 1, 3, 1)
    y = torch.tensor([1., -1., 0.], device=default_device).reshape(1, 1, 3, 1)
    _, _, cx, cy, ray_ok, _ = tr.trace_rays(specs, lens, xy=(x, y))
    axis = torch.clamp(2 * cy[:, 0, 0, 0] ** 2, min=1e-6)
    ri = (cy[..., 0, :] - cy[..., 1, :]) * cx[..., 2, :] / axis[:, None, None]
    valid = ray_ok.all(dim=2)                                   # [lens, field, wavelength]
    valid = valid & valid[:, :1, :]
    return torch.where(valid, ri, torch.ones_like(ri))


def compute_ray_aiming_error(specs, lens, rel_fields, vig_fn=None, n_ray_aiming_iter=1, ray_aiming_mode='real',
                             default_device="cuda"):
    """Relative error, at the aperture stop, of the upper and lower meridional rays after ray aiming:
    y_stop / r_stop - y_pupil (0 for a lens whose stop is its first row)."""
    specs2, lens2 = specs.up_to_stop(), lens.up_to_stop()
    if (lens2.structure.stop_idx == 0).all():
        return 0
    if ray_aiming_mode == 'paraxial':
        rs = (paraxial.compute_magnification(lens2) * specs2.epd / 2).reshape(-1, 1, 1, 1)
    elif ray_aiming_mode == 'real':
        rs = paraxial.compute_pupil_radius(specs2, lens2, default_device=default_device).reshape(-1, 1, 1, 1)
    else:
        raise ValueError("ray_aiming_mode must be 'real' or 'paraxial'")
    y = torch.tensor([-1., 1.], device=default_device).reshape(1, 1, 2, 1)
    x = torch.zeros_like(y)
    tr = RayTracer(mode='tee', rel_fields=rel_fields, vig_fn=vig_fn, wavelengths=['d'],
                   n_ray_aiming_iter=n_ray_aiming_iter, ray_aiming_mode=ray_aiming_mode, default_device=default_device)
    # the truncated lens ends at the stop: its "image plane" is the stop plane, and ray aiming (computed on
    # the same rows) is applied by trace_rays itself
    _, ys, *_ = tr.trace_rays(specs2, lens2, xy=(x, y), use_vig=True)
    if vig_fn is not None:
        fields = torch.tensor(list(rel_fields), dtype=torch.float32, device=default_device)[None, :]
        y = apply_vignetting(y, vig_fn(fields, specs.vig_up), vig_fn(fields, specs.vig_down))
    return ys / rs - y


def compute_psf(x, y, n_bins=(21, 21), increment=None, y_target=None, weights=None, y_extent="reference"):
    """Soft-histogram PSF of a ray fan on a per-field pixel grid (the reference's TensorFlow
    `compute_psf`, ray_tracing.py:206-270; unreachable in its PyTorch port, PARITY UNPINNED).

    x, y: [n_lens, n_fields, n_channels, n_rays] image-plane coordinates (`psf_from_trace` brings the
    tracer's [1,F,P,W] outputs into this layout).  One grid per (lens, field), centred at x = 0 and
    y = y_target (default: the mean y of the field); pixel size `increment`, or the fan's extent / n_bins.
    Every ray adds a Gaussian of sigma = half a pixel to each pixel centre; as in the reference only the
    x >= 0 half of the grid is evaluated and mirrored (the pupil is sampled symmetrically in x), and each
    channel's kernel is normalised to unit sum.
    `weights` (extension; same shape as x): per-ray weights, e.g. ray_ok as float to leave failed rays out.
    `y_extent` (increment=None only): "reference" (default) sizes the grid exactly as the reference text does --
    y_size = 2 max(y_max - y_target, y_target - y_min) with y_min, y_max taken AFTER y was centred on y_target
    (ray_tracing.py:220-231), i.e. y_target is subtracted twice (sic): for an off-axis field (y_target = 3 mm, spot
    +-0.06 mm) the grid spans 6.12 mm, not 0.12 mm, and the spot fills one pixel.  Kept because this function stands in
    for that text number for number; "centred" is the evident intent, y_size = 2 max(y_max, -y_min) of the centred y
    (a deviation from the reference, off by default; `psf_from_trace`, an extension, uses it).

    Returns (x_size, y_size, y_target, kernels [n_grids, n_channels, n_y_bins, n_x_bins],
    accounted_ray_proportion [n_grids]).

    The double sum over rays and pixels is a contraction over the ray index, so it is evaluated as one batched
    GEMM  G_y [ny, R] @ G_x^T [R, nx]  per (grid, channel) instead of the reference's [.., ny, nx, R] tensor.
    """
    nw, nr = x.shape[-2], x.shape[-1]
    n_grids = x.shape[0] * x.shape[1]
    n_x_bins, n_y_bins = n_bins
    x = x.reshape(n_grids, nw, nr)
    y = y.reshape(n_grids, nw, nr)
    if y_target is None:
        y_target = y.reshape(n_grids, -1).mean(dim=1)
    y = y - y_target[:, None, None]
    if increment is not None:
        x_incr = y_incr = torch.ones(n_grids, dtype=x.dtype, device=x.device) * increment
        x_size = increment * n_x_bins
        y_size = increment * n_x_bins                  # (sic: the reference uses n_x_bins for both)
    else:
        flat_y = y.reshape(n_grids, -1)
        y_c_min, y_c_max = flat_y.min(dim=1).values, flat_y.max(dim=1).values      # y is centred already
        x_size = x.reshape(n_grids, -1).max(dim=1).values
        if y_extent == "reference":
            y_size = 2 * torch.maximum(y_c_max - y_target, y_target - y_c_min)     # (sic) ray_tracing.py:231
        elif y_extent == "centred":
            y_size = 2 * torch.maximum(y_c_max, -y_c_min)
        else:
            raise ValueError("y_extent must be 'reference' or 'centred'")
        x_incr = x_size / n_x_bins
        y_incr = y_size / n_y_bins
    rng = lambda n: torch.arange(n, dtype=x.dtype, device=x.device)                 # noqa: E731
    if n_x_bins % 2 == 1:
        gx = rng(n_x_bins // 2 + 1)[None, :] * x_incr[:, None]
    else:
        gx = (rng(n_x_bins // 2) + 0.5)[None, :] * x_incr[:, None]
    gy = (rng(n_y_bins) + 0.5 - n_y_bins / 2)[None, :] * y_incr[:, None]
    sig_x, sig_y = (x_incr / 2)[:, None, None, None], (y_incr / 2)[:, None, None, None]
    g_x = torch.exp(-((x[:, :, None, :] - gx[:, None, :, None]) / sig_x) ** 2 / 2)   # [g, w, nx_half, r]
    g_y = torch.exp(-((y[:, :, None, :] - gy[:, None, :, None]) / sig_y) ** 2 / 2)   # [g, w, ny, r]
    if weights is not None:
        g_y = g_y * weights.reshape(n_grids, nw, 1, nr).to(g_y.dtype)
    kernels = torch.matmul(g_y, g_x.transpose(-1, -2))                               # [g, w, ny, nx_half]
    if n_x_bins % 2 == 1:
        kernels = torch.cat((torch.flip(kernels[..., 1:], dims=(-1,)), kernels), dim=-1)
    else:
        kernels = torch.cat((torch.flip(kernels, dims=(-1,)), kernels), dim=-1)
    kernels = kernels / kernels.sum(dim=(-1, -2), keepdim=True)
    xs = x_size if torch.is_tensor(x_size) else torch.full((n_grids,), float(x_size), dtype=x.dtype, device=x.device)
    ys = y_size if torch.is_tensor(y_size) else torch.full((n_grids,), float(y_size), dtype=x.dtype, device=x.device)
    accounted = (y.abs() < ys[:, None, None] / 2) & (x.abs() < xs[:, None, None] / 2)
    return x_size, y_size, y_target, kernels, accounted.to(x.dtype).mean(dim=(-1, -2))


def psf_from_trace(x, y, ray_ok=None, n_bins=(21, 21), increment=None, y_target=None, y_extent="centred"):
    """compute_psf on the outputs of RayTracer.trace_rays / trace_skew ([1, F, P, W]; their memory is already
    [F, W, P], so the permutation below is free).  With `ray_ok` failed rays are left out of the histogram and of
    the default y_target (the reference counts them as rays at the origin).  An extension with no counterpart in the
    reference: the automatic grid is sized on the centred spot (`y_extent="centred"`, see compute_psf)."""
    xt, yt = x.permute(0, 1, 3, 2), y.permute(0, 1, 3, 2)
    w = None
    if ray_ok is not None:
        w = ray_ok.permute(0, 1, 3, 2).to(y.dtype)
        if y_target is None:
            n_g = yt.shape[0] * yt.shape[1]
            y_target = (yt * w).reshape(n_g, -1).sum(dim=1) / w.reshape(n_g, -1).sum(dim=1).clamp_min(1)
    return compute_psf(xt, yt, n_bins=n_bins, increment=increment, y_target=y_target, weights=w, y_extent=y_extent)
