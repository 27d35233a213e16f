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
