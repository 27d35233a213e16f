"""
Paraxial (ABCD ray-transfer matrix) utilities: the step BEFORE the hot path.

They produce the pupil position `z` fed to `trace_skew`, the EFL/BFL and the last curvature.
O(rows) host-side work on tiny tensors: stays in PyTorch so autograd chains through it.
Mirrors ray_tracing_lite.py:301-350 (`reduce_abcd`, `interface_propagation_abcd`,
`compute_pupil_position`), :725-794 (`compute_last_curvature`, `get_first_order`) and
:834-844 (`compute_pupil_radius`).  The pairwise product order of `reduce_abcd` is kept
because it fixes the fp32 value of `z`.
"""
from __future__ import annotations

import torch

from .lens_modeling import Lens, mask_replace


def interface_propagation_abcd(c: torch.Tensor, t: torch.Tensor, n: torch.Tensor) -> torch.Tensor:
    """Per row: refraction at curvature c (index n_k -> n_k+1) followed by a gap t.

    c, t: [lens, rows]; n: [lens, rows+1].  Returns [lens, rows, 2, 2] = [[A, B], [C, D]].
    """
    assert n.shape[-1] - 1 == c.shape[-1] == t.shape[-1]
    ratio = n[:, :-1] / n[:, 1:]
    power = c * (ratio - 1)
    m = torch.stack((1 + power * t, ratio * t, power, ratio), dim=-1)
    return m.reshape(n.shape[0], -1, 2, 2)


def reduce_abcd(abcd: torch.Tensor) -> torch.Tensor:
    """Ordered product M_last ... M_1 M_0 by pairwise halving; returns [lens, 2, 2]."""
    while abcd.shape[1] > 1:
        n = abcd.shape[1]
        even = n - (n % 2)
        pairs = abcd[:, 1:even:2] @ abcd[:, 0:even:2]
        abcd = pairs if n % 2 == 0 else torch.cat((pairs, abcd[:, -1:]), dim=1)
    return abcd.squeeze(1)


def _with_air_in_front(nd: torch.Tensor) -> torch.Tensor:
    return torch.cat((torch.ones_like(nd[:, :1]), nd), dim=1)


def compute_pupil_position(lens: Lens, mode=None, front: Lens = None) -> torch.Tensor:
    """Paraxial entrance-pupil position w.r.t. the first vertex: B/A of the rows before the stop.
    `mode` (GPU only): 'strict' = the reference's fp32 value bit for bit, 'fast' = fp64 inside, rounded once;
    default ops.get_default_mode().  `front`: lens.up_to_stop() if the caller has it already."""
    if front is None:
        front = lens.up_to_stop()
    if front.structure.mask.shape[1] == 0:
        return torch.zeros(len(front), dtype=lens.c.dtype, device=lens.c.device)
    from .lens_modeling import _memoised
    n = _memoised("n_front", (), (front.nd,), (front.nd,), lambda: _with_air_in_front(front.nd))     # constant glasses: once
    if front.c.is_cuda and front.c.dtype == torch.float32:
        # on the GPU (the hot configuration): the whole chain below and its autograd backward are one tiny kernel
        # each (tl_pupil_position, one thread per lens, fp64 inside) instead of ~25 + ~60 launches
        from . import ops
        return ops.pupil_position(front.c, front.t, n, mode).to(lens.c.dtype)
    m = reduce_abcd(interface_propagation_abcd(front.c, front.t, n))
    return m[:, 0, 1] / m[:, 0, 0]


def get_first_order(lens: Lens):
    """(EFL, BFL) from the system matrix with the last gap removed."""
    dev = lens.structure.mask_torch.device
    rows = torch.arange(len(lens), device=dev)
    last = lens.structure.mask_torch.sum(dim=1) - 1
    t = lens.t.clone()
    t[rows, last] = torch.zeros(len(lens), dtype=t.dtype, device=t.device)
    m = reduce_abcd(interface_propagation_abcd(lens.c, t, _with_air_in_front(lens.nd)))
    return -1 / m[:, 1, 0], -m[:, 0, 0] / m[:, 1, 0]


def compute_magnification(lens: Lens) -> torch.Tensor:
    """First-order magnification (A element); used by ray_aiming_mode='paraxial'."""
    m = reduce_abcd(interface_propagation_abcd(lens.c, lens.t, _with_air_in_front(lens.nd)))
    return m[:, 0, 0]


def compute_last_curvature(structures, c: torch.Tensor, t: torch.Tensor, nd: torch.Tensor) -> torch.Tensor:
    """Solve the last glass/air curvature so that EFL = 1 (ray_tracing_lite.py:725-769).

    c: flat curvatures WITHOUT the last row; t: flat thicknesses; nd: flat glass indices.
    Returns the flat curvature vector including the solved one.
    """
    mask = structures.mask_torch
    dev = mask.device
    n_lens = mask.shape[0]
    rows = torch.arange(n_lens, device=dev)
    length = mask.sum(dim=1)
    ends_air_air = ~structures.mask_G_torch[rows, length - 2]
    solve_at = length - 1 - ends_air_air.long()

    is_last = torch.zeros_like(mask).scatter_(1, (length - 1)[:, None], True)
    c_rows = mask & ~is_last
    c2d = mask_replace(c_rows.cpu().numpy(), torch.zeros(mask.shape, dtype=c.dtype, device=dev), c)
    t2d = mask_replace(structures.mask, torch.zeros(mask.shape, dtype=t.dtype, device=dev), t)
    n2d = mask_replace(structures.mask_G, torch.ones(mask.shape, dtype=nd.dtype, device=dev), nd)
    n2d = _with_air_in_front(n2d)

    is_solved = torch.zeros_like(mask).scatter_(1, solve_at[:, None], True)
    use = (c_rows & ~is_solved)[..., None, None]
    abcd = interface_propagation_abcd(c2d, t2d, n2d)
    abcd = torch.where(use.expand_as(abcd), abcd, torch.eye(2, dtype=abcd.dtype, device=dev)[None, None])
    m = reduce_abcd(abcd)

    n_before = n2d[rows, solve_at]
    last_c = -(1 + n_before * m[:, 1, 0]) / (m[:, 0, 0] * (n_before - 1))
    c2d = c2d.clone().scatter_(1, solve_at[:, None], last_c[:, None])
    return c2d[mask]


def compute_pupil_radius(specs, lens2stop, default_device="cuda") -> torch.Tensor:
    """Height of the marginal ray (relative pupil y = 1, on axis, d line) at the stop."""
    from .ray_tracing import RayTracer
    x = torch.zeros(1, 1, 1, 1, device=default_device)
    y = torch.ones(1, 1, 1, 1, device=default_device)
    tracer = RayTracer(rel_fields=[0.], vig_fn=None, wavelengths=['d'], default_device=default_device)
    _, yp, *_ = tracer.trace_rays(specs, lens2stop, xy=(x, y), use_vig=False)
    return yp.reshape(yp.shape[0])
