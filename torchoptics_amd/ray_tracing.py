"""
Sequential ray tracer: drop-in for the reference's `torchlens/ray_tracing_lite.py`.

Public names and signatures follow the reference (`RayTracer` ray_tracing_lite.py:26-208,
`trace_skew` :594-675, `compute_rms2d` :678-702, samplers :353-422, `scale_to_epd` :497-507),
but the per-surface Python loop of ~86 eager tensor ops is gone: `trace_skew` is ONE fused HIP
kernel (and one hand-written backward kernel) reached through the C ABI in
include/tl_trace.h.  There is no CPU implementation in this package: CPU tensors raise.

Tensor dims, as in the reference: [lens (=1), field, pupil, wavelength(, surface)].
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import ops
from .paraxial import (compute_last_curvature, compute_magnification, compute_pupil_position,  # noqa: F401
                       compute_pupil_radius, get_first_order, interface_propagation_abcd, reduce_abcd)

_LINES = {'C': 656.3, 'd': 587.6, 'F': 486.1}
# ray aiming through the fused kernel tl_ray_aim (GPU, fp32 lenses, 'real' mode, no vignetting function); False: the
# reference's sequence of tensor ops (two traces + autograd), kept for CPU tensors, the other modes and as the checker
_AIM_KERNEL = True


def set_ray_aiming_kernel(on: bool) -> None:
    global _AIM_KERNEL
    _AIM_KERNEL = bool(on)
_MODES = ('skew_random', 'tee', 'circular', 'skew_uniform_half_equidistant', 'skew_uniform_half_jittered',
          'skew_inner_square_half', 'skew_outer_edge_uniform', 'meridional_uniform', 'sagittal_uniform', 'chief')


# ---------------------------------------------------------------------------- pupil samplers
def tee(tensor=None, device="cuda"):
    """Bottom and top meridional rays and the +x sagittal ray, shape [1,1,3,1]."""
    # cached constants: no host-to-device copy per call, so ray aiming can sit inside a captured HIP graph
    from .lens_modeling import const_tensor
    y = const_tensor([-1., 1., 0.], torch.float32, device, (1, 1, 3, 1))
    x = const_tensor([0., 0., 1.], torch.float32, device, (1, 1, 3, 1))
    return x, y


def _cos_sin(th):
    """cos and sin of the fp32 grid angles.  On the GPU: evaluated in fp64 and rounded once, i.e. the correctly rounded
    fp32 values -- what the reference's host libm gives wherever it rounds correctly -- rather than the device's fp32
    cos / sin (1-2 ulp), so that the fan a GPU run traces is the reference's fan to the last bit on almost every ray."""
    if th.is_cuda and th.dtype == torch.float32:
        t64 = th.double()
        return torch.cos(t64).float(), torch.sin(t64).float()
    return torch.cos(th), torch.sin(th)


def circle(tensor, n_r, n_theta, default_device="cuda"):
    """Polar grid: radii linspace(0,1,n_r) and angles linspace(0,2pi,n_theta), both without the
    end point -- so n_theta coincident rays at r=0 and none at r=1 (reference quirk B7, kept).
    Small grids are cached per device and returned as shared tensors: treat them as read-only."""
    # the grid is a constant of (n_r, n_theta, device): small ones are kept, so that a step that builds its fan inside
    # a captured HIP graph does no host-to-device copy (and an eager loop saves six launches per step)
    dev = torch.device(default_device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())       # "cuda" means another device after set_device
    key = (n_r, n_theta, dev)
    hit = _CIRCLE.get(key)
    if hit is not None:
        return hit          # shared, READ-ONLY tensors (the reference returns fresh ones: clone before editing in place)
    r = torch.from_numpy(np.linspace(0, 1.0, n_r, endpoint=False, dtype=np.float32)).to(default_device)
    th = torch.from_numpy(np.linspace(0, 2 * np.pi, n_theta, endpoint=False, dtype=np.float32)).to(default_device)
    cos_th, sin_th = _cos_sin(th)
    x = r[None, :, None] * cos_th[None, None, :]
    y = r[None, :, None] * sin_th[None, None, :]
    out = x.reshape(-1, 1, n_r * n_theta, 1), y.reshape(-1, 1, n_r * n_theta, 1)
    if n_r * n_theta <= (1 << 16) and len(_CIRCLE) < 16:
        _CIRCLE[key] = out
    return out


_CIRCLE = {}


def circle_index_range(n_r, n_theta, start, stop, default_device="cuda"):
    """Points [start, stop) of `circle`'s grid generated from their indices (pupil sharding:
    every rank builds its own contiguous slice; bit-identical to slicing the full grid)."""
    r_all = torch.from_numpy(np.linspace(0, 1.0, n_r, endpoint=False, dtype=np.float32)).to(default_device)
    th_all = torch.from_numpy(np.linspace(0, 2 * np.pi, n_theta, endpoint=False, dtype=np.float32)).to(default_device)
    idx = torch.arange(start, stop, device=default_device)
    cos_all, sin_all = _cos_sin(th_all)
    r, k = r_all[idx // n_theta], idx % n_theta
    return (r * cos_all[k]).reshape(1, 1, -1, 1), (r * sin_all[k]).reshape(1, 1, -1, 1)


def circle_pseudo_random(tensor, n_r, n_theta):
    """Stratified jitter over the unit disc; draws from the global CPU torch RNG like the
    reference (two torch.rand calls of shape [numel, n_r, n_theta])."""
    n_el = int(np.prod(tensor.shape))
    d_r2 = torch.rand((n_el, n_r, n_theta)) / n_r
    d_th = torch.rand((n_el, n_r, n_theta)) / n_theta
    r2_0 = torch.tensor(np.linspace(0, 1, n_r, endpoint=False, dtype=np.float32))[None, :, None]
    th_0 = torch.tensor(np.linspace(0, 1, n_theta, endpoint=False, dtype=np.float32))[None, None, :]
    r = torch.sqrt(d_r2 + r2_0)
    th = (d_th + th_0) * 2 * np.pi
    n = n_r * n_theta
    return (r * torch.cos(th)).view(-1, 1, n, 1), (r * torch.sin(th)).view(-1, 1, n, 1)


# The remaining sampling modes exist in the reference only as commented-out TensorFlow code
# (ray_tracing_lite.py:363-480; originals in ray_tracing.py:358-476) and raise NameError in its PyTorch
# port.  Written here from their stated intent; PARITY UNPINNED (the TF original cannot run here).
def _as_rays(x, y, device):
    x = torch.as_tensor(np.asarray(x, dtype=np.float32)).to(device)
    y = torch.as_tensor(np.asarray(y, dtype=np.float32)).to(device)
    return x.reshape(1, 1, -1, 1), y.reshape(1, 1, -1, 1)


def chief(tensor, _n=None, device="cuda"):
    """The chief ray: pupil centre."""
    return _as_rays([0.0], [0.0], device)


def meridional_uniform(tensor, n_rays, device="cuda"):
    """n_rays points on the y axis of the pupil, from -1 to 1 inclusive."""
    y = np.linspace(-1.0, 1.0, n_rays)
    return _as_rays(np.zeros_like(y), y, device)


def sagittal_uniform(tensor, n_rays, device="cuda"):
    """n_rays points on the positive x axis of the pupil, from 0 to 1 inclusive."""
    x = np.linspace(0.0, 1.0, n_rays)
    return _as_rays(x, np.zeros_like(x), device)


def circle_outer_edge_uniform(tensor, n_rays, device="cuda"):
    """n_rays points on the rim of the pupil, equally spaced in angle starting at +x."""
    th = np.linspace(0, 2 * np.pi, n_rays, endpoint=False, dtype=np.float32)
    return _as_rays(np.cos(th), np.sin(th), device)


def _half_shells(n_r, n_i):
    """Shell index of every ray and its polar angle for the right-half-pupil patterns: shell i holds
    n_i (2 i + 1) rays spread over (-pi/2, pi/2) at the midpoints of equal angular bins (equal area per ray)."""
    counts = [n_i * (2 * i + 1) for i in range(n_r)]
    shell = np.repeat(np.arange(n_r), counts)
    theta = np.concatenate([((np.arange(n) + 0.5) / n - 0.5) * np.pi for n in counts])
    return shell, theta


def skew_uniform_half_equidistant(tensor, n_r, n_i, device="cuda"):
    """n_i n_r^2 rays over the right half of the pupil; shell i sits at radius (i + 0.5) / n_r."""
    shell, theta = _half_shells(n_r, n_i)
    r = (shell + 0.5) / n_r
    return _as_rays(r * np.cos(theta), r * np.sin(theta), device)


def skew_uniform_half_jittered(tensor, n_r, n_i, device="cuda"):
    """As above, but the rays of a shell alternate between its inner radius and the next half step
    outwards (radii 0, 1/(2 n_r - 1), 2/(2 n_r - 1), ... 1), so that the pupil edge is sampled."""
    shell, theta = _half_shells(n_r, n_i)
    inner = np.linspace(0, 1, 2 * n_r)[::2]
    step = 1.0 / (2 * n_r - 1)
    r = inner[shell] + step * ((np.arange(len(shell)) + shell) % 2)
    return _as_rays(r * np.cos(theta), r * np.sin(theta), device)


def skew_inner_square_half(tensor, n_y, _unused=None, device="cuda"):
    """n_y x n_y grid over the right half of the square inscribed in the pupil."""
    xs = np.linspace(-1, 1, 2 * n_y)[-n_y:] / np.sqrt(2)
    ys = np.linspace(-1, 1, n_y) / np.sqrt(2)
    gx, gy = np.meshgrid(xs, ys)          # rows: y, columns: x
    return _as_rays(gx.ravel(), gy.ravel(), device)


def apply_vignetting(y, vig_up, vig_down):
    """Squeeze relative pupil coordinates: the upper edge moves in by vig_up, the lower by vig_down."""
    pad = [1] * (y.dim() - vig_down.dim())
    vig_up = vig_up.reshape(*vig_up.shape, *pad)
    vig_down = vig_down.reshape(*vig_down.shape, *pad)
    return y * (1 - (vig_up + vig_down) / 2) + (vig_down - vig_up) / 2


def scale_to_epd(y, epd):
    """Relative pupil coordinate -> height at the pupil plane: y * epd / 2."""
    if y.dim() == 4 and not epd.requires_grad:
        # (y epd) / 2 == y (epd / 2) bit for bit (a power of two); epd / 2 is the same tensor every step
        from .lens_modeling import _memoised
        return y * _memoised("epd_half", (), (epd,), (epd,), lambda: epd.reshape(-1, 1, 1, 1) / 2)
    return y * epd.reshape(-1, *([1] * (y.dim() - 1))) / 2


# ---------------------------------------------------------------------------- the hot path
def _as_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not torch.is_tensor(t):
        raise TypeError(f"{name} must be a tensor")
    return t if t.dtype == torch.float32 else t.to(torch.float32)


def _as_f64(t: torch.Tensor, name: str) -> torch.Tensor:
    if not torch.is_tensor(t):
        raise TypeError(f"{name} must be a tensor")
    return t if t.dtype == torch.float64 else t.to(torch.float64)


def _dense(a: torch.Tensor) -> torch.Tensor:
    return a if a.is_contiguous() else a.contiguous()


def _rows(a: torch.Tensor, n_lens: int, tail: tuple) -> torch.Tensor:
    """[n_lens, *tail] contiguous from an argument that holds one set of values per lens, or one set for all."""
    n_tail = 1
    for d in tail:
        n_tail *= d
    if a.numel() == n_lens * n_tail:
        return _dense(a.reshape(n_lens, *tail))
    return _dense(a.reshape(1, *tail).expand(n_lens, *tail))


def trace_skew(x, y, z, cx, cy, c, t, mu, mask, aggregate=False, allow_backward_rays=True, mode=None,
               want_rays=True, kappa=None, poly=None, surf_kind=None, n_index=None, want_opd=False, x_moments=False):
    """Trace rays from the entrance pupil to the image plane through S surface rows.

    Same contract as the reference (ray_tracing_lite.py:594-675): inputs broadcast to
    [B, F, P, W]; returns (x, y, cx, cy, ray_ok, ray_backward).  Differentiable w.r.t.
    x, y, z, cx, cy, c, t, mu through a hand-written backward kernel.  A batch of B > 1 padded lenses
    (every argument's dim 0 is 1 or B) is ONE kernel launch each way; padded rows (c = 0, t = 0, mu = 1,
    mask False) are traced as the identity rows they are in the reference.

    Extras (not in the reference):
      aggregate   True: a 7th return value `stacks` = {'z_RELU', 'theta_norm', 'theta_prime_norm'}, each a
                  list of S tensors [1,F,P,W] as in the reference (:641-657).  The stacks are plain values;
                  the differentiable quantity is their fused sum, see `penalty_sum(stacks, n_sequence)`.
                  'sum': the same 7th value WITHOUT the per-surface tensors -- only the fused sums that
                  `penalty_sum` / `unsupervised_loss` read (the lists are 12 S bytes of HBM writes per ray);
      mode        'strict' | 'fast' arithmetic (default ops.get_default_mode());
      kappa, poly aspheric rows: conic constants [S] and even polynomial terms [S,4] (a4..a10);
                  `surf_kind` [S] (bool/int) marks the rows traced by Newton iteration -- default: the
                  rows where kappa or poly is non-zero.  Differentiable w.r.t. kappa and poly too;
      n_index, want_opd   refractive indices [1,1,1,W,S+1] (entry 0 = object space) and a seventh
                  return value: the optical path length per ray, differentiable w.r.t. every lens and launch
                  parameter and w.r.t. n_index (its backward runs the checkpoint kernel);
      x_moments   also fuse the x-moments of the spot into the trace (for `compute_rms_spot_xy`; compute_rms2d reads
                  y only, ray_tracing_lite.py:684-701, so by default the kernel does not spend time on them);
      the returned `y` carries the fused spot moments so `compute_rms2d(x, y, ray_ok)` costs no
      second pass over the rays.
    """
    B = max(a.shape[0] for a in (x, y, z, cx, cy, c, t, mu, mask) if torch.is_tensor(a) and a.dim() >= 4)
    if B > 1:
        # one launch takes at most _MAX_GRID_ROWS (lens, field, wavelength) rows (the grid's y dimension); the
        # reference's broadcasting has no such bound (4 096 lenses x 8 fields x 3 wavelengths = 98 304 rows): trace a
        # larger batch in lens chunks and join the results
        rows_per_lens = (max(a.shape[1] for a in (x, y, cx, cy)) * max(x.shape[3], y.shape[3], mu.shape[3]))
        if B * rows_per_lens > _MAX_GRID_ROWS:
            return _trace_skew_in_lens_chunks(max(1, _MAX_GRID_ROWS // rows_per_lens), B, x, y, z, cx, cy, c, t, mu, mask,
                                              aggregate, allow_backward_rays, mode, want_rays, kappa, poly, surf_kind,
                                              n_index, want_opd, x_moments)
    # double precision (RayTracer(double_precision=True)): the generic fp64 kernels, through the Python host chain below
    f64 = any(torch.is_tensor(a) and a.dtype == torch.float64 for a in (x, y, z, cy, c, t, mu))
    if f64 and (aggregate or want_opd):
        raise NotImplementedError("the double-precision trace has no penalty term and no optical path length")
    ext = None if f64 else ops._ext()
    if ext is not None:
        # the C++ host chain: shapes are normalised, outputs allocated and the autograd node built in csrc/tl_torch.cpp
        S = c.shape[-1]
        kap = pol = kind = None
        if kappa is not None or poly is not None:
            kap = kappa if kappa is not None else torch.zeros(1, S, device=c.device)
            pol = poly if poly is not None else torch.zeros(1, S, 4, device=c.device)
            kind = surf_kind
            if kind is None:
                kind = (kap.detach().reshape(-1, S) != 0) | (pol.detach().reshape(-1, S, 4) != 0).any(dim=-1)
            kind = torch.as_tensor(kind, device=c.device)
        out, use_inv = ops.trace_cpp(ext, x, y, z, cx, cy, c, t, mu, mask, kap, pol, kind, n_index if want_opd else None,
                                     bool(allow_backward_rays), mode or ops.get_default_mode(), want_rays, bool(want_opd),
                                     bool(aggregate), bool(aggregate is True and want_rays), bool(x_moments))
        n_pw = max(x.shape[2], y.shape[2]) * max(x.shape[3], y.shape[3], mu.shape[3])
        return _trace_result(out, use_inv, want_rays, want_opd, aggregate, x_moments, n_pw, B)
    cast = _as_f64 if f64 else _as_f32          # (the normalisation below is dtype-blind)
    x, y, z, cx, cy = (cast(a, n) for a, n in ((x, 'x'), (y, 'y'), (z, 'z'), (cx, 'cx'), (cy, 'cy')))
    c, t, mu = cast(c, 'c'), cast(t, 't'), cast(mu, 'mu')
    for a, n in ((x, 'x'), (y, 'y'), (cx, 'cx'), (cy, 'cy'), (z, 'z')):
        if a.dim() != 4 or a.shape[0] not in (1, B):
            raise ValueError(f"{n} must be 4-D with 1 or {B} lenses in dim 0, got {tuple(a.shape)}")
    if c.dim() != 5 or t.dim() != 5 or mu.dim() != 5 or any(a.shape[0] not in (1, B) for a in (c, t, mu)):
        raise ValueError("c, t, mu must be 5-D [1|B,1,1,1|W,S]")
    S = c.shape[-1]
    F = max(x.shape[1], y.shape[1], cx.shape[1], cy.shape[1])
    P = max(x.shape[2], y.shape[2])
    W = max(x.shape[3], y.shape[3], mu.shape[3])
    if z.numel() not in (1, B):
        raise ValueError("z must hold one pupil position per lens")
    x_e, y_e = x.expand(B, F, P, W), y.expand(B, F, P, W)
    if any(a.shape[1] not in (1, F) or a.shape[2] != 1 or a.shape[3] != 1 for a in (cx, cy)):
        raise ValueError("cx, cy must be per-field [1|B,1|F,1,1]")
    # (one view op per differentiable argument where the shapes allow: every reshape / expand of a leaf is a node the
    #  autograd engine walks on the way back, ~5-10 us of host time each)
    cx2, cy2 = _dense(cx.reshape(cx.shape[0], cx.shape[1])), _dense(cy.reshape(cy.shape[0], cy.shape[1]))
    zv = _rows(z, B, ())
    c2, t2 = _rows(c, B, (S,)), _rows(t, B, (S,))
    mu3 = _rows(mu, B, (W, S)) if mu.shape[3] == W else _dense(mu.reshape(mu.shape[0], 1, S).expand(B, W, S))
    mask_u8 = mask.reshape(-1, S)
    mask_u8 = mask_u8.view(torch.uint8) if mask_u8.dtype == torch.bool else mask_u8.to(torch.uint8)   # bool: no copy
    mask_u8 = mask_u8.expand(B, S).contiguous()
    kap = pol = kind_u8 = None
    if kappa is not None or poly is not None:
        # per lens [B,S] / [B,S,4], or one set [S] / [S,4] shared by every lens of the batch
        kap = (cast(kappa, 'kappa').reshape(-1, S) if kappa is not None
               else torch.zeros(1, S, device=c.device, dtype=c.dtype)).expand(B, S).contiguous()
        pol = (cast(poly, 'poly').reshape(-1, S, 4) if poly is not None
               else torch.zeros(1, S, 4, device=c.device, dtype=c.dtype)).expand(B, S, 4).contiguous()
        if surf_kind is None:
            surf_kind = (kap.detach() != 0) | (pol.detach() != 0).any(dim=-1)
        kind_u8 = torch.as_tensor(surf_kind, device=c.device).reshape(-1, S).to(torch.uint8).expand(B, S).contiguous()
    nidx = None
    if want_opd:
        if n_index is None:
            raise ValueError("want_opd=True needs n_index [1|B,1,1,W,S+1]")
        nidx = cast(n_index, 'n_index')
        nidx = nidx.reshape(nidx.shape[0] if nidx.dim() == 5 else 1, -1, S + 1).expand(B, W, S + 1).contiguous()
    if f64:
        out = ops.TraceFunctionF64.apply(x_e, y_e, zv, cx2, cy2, c2, t2, mu3, kap, pol, mask_u8, kind_u8,
                                         bool(allow_backward_rays), want_rays)
        return _trace_result((*out, None, None), False, want_rays, False, False, True, P * W, B)
    out = ops.TraceFunction.apply(x_e, y_e, zv, cx2, cy2, c2, t2, mu3, kap, pol, mask_u8, kind_u8, nidx,
                                  bool(allow_backward_rays), mode or ops.get_default_mode(), want_rays, bool(want_opd),
                                  bool(aggregate), bool(aggregate is True and want_rays), bool(x_moments))
    return _trace_result(out, ops._last_use_inv, want_rays, want_opd, aggregate, x_moments, P * W, B)


def _trace_result(out, use_inv, want_rays, want_opd, aggregate, x_moments, n_pw, B):
    """The tuple trace_skew returns, from the nine outputs of the trace function."""
    xo, yo, cxo, cyo, ok, back, moments, opd, stk = out[:9]
    rms = out[9] if len(out) > 9 and out[9].numel() > 0 else None      # C++ host chain: the spot metric, fused into the trace node
    if want_rays:
        # remember which moments belong to these rays (checked by identity + version in compute_rms2d):
        # [B*F, TL_NMOM], lens-major; and their spot metric when the trace node computed it already
        yo._tl_spot = (moments, ok, yo._version, n_pw, bool(x_moments), rms)
        xo._tl_use_inv = bool(use_inv)                 # which backward algorithm this trace will take (ops.used_walk_back)
        res = (xo, yo, cxo, cyo, ok, back)
        if want_opd:
            res += (opd,)
        if aggregate:
            res += (PenaltyStacks(stk if aggregate is True else None, moments, B),)
        return res
    return moments


_MAX_GRID_ROWS = 65535          # gridDim.y limit = B * F * W of one launch (tl_trace.h: tl_problem.B)


def _trace_skew_in_lens_chunks(nb, B, x, y, z, cx, cy, c, t, mu, mask, aggregate, allow_backward_rays, mode, want_rays,
                               kappa, poly, surf_kind, n_index, want_opd, x_moments):
    """trace_skew of a lens batch too large for one launch: `nb` lenses at a time, results joined along the lens axis
    (the per-ray outputs are copied once more -- such batches are many small fans, not few large ones)."""
    def part(a, b0, b1, lens_dims):
        # an argument that carries the lens axis (dim 0 of size B) is sliced, one shared by all lenses is passed as is
        if not torch.is_tensor(a) or a.dim() not in lens_dims or a.shape[0] != B:
            return a
        return a[b0:b1]
    outs = []
    for b0 in range(0, B, nb):
        b1 = min(B, b0 + nb)
        args = [part(a, b0, b1, (4, 5)) for a in (x, y, z, cx, cy, c, t, mu, mask)]
        kw = dict(kappa=part(kappa, b0, b1, (2,)), poly=part(poly, b0, b1, (3,)),
                  surf_kind=part(surf_kind, b0, b1, (2,)) if torch.is_tensor(surf_kind) else surf_kind,
                  n_index=part(n_index, b0, b1, (5,)))
        outs.append(trace_skew(*args, aggregate, allow_backward_rays, mode=mode, want_rays=want_rays, want_opd=want_opd,
                               x_moments=x_moments, **kw))
    if not want_rays:
        return torch.cat(outs, dim=0)                      # moments [B*F, TL_NMOM], lens-major
    n_ray = 6 + (1 if want_opd else 0)
    res = [torch.cat([o[i] for o in outs], dim=0) for i in range(n_ray)]
    moments = torch.cat([o[1]._tl_spot[0] for o in outs], dim=0)
    res[1]._tl_spot = (moments, res[4], res[1]._version, outs[0][1]._tl_spot[3], bool(x_moments))
    if aggregate:
        stk = None
        if aggregate is True:
            keys = ('z_RELU', 'theta_norm', 'theta_prime_norm')
            stk = torch.stack([torch.stack([torch.cat([o[n_ray][k][j] for o in outs], dim=0)
                                            for j in range(len(outs[0][n_ray][k]))]) for k in keys])
        res.append(PenaltyStacks(stk, moments, B))
    return tuple(res)


class PenaltyStacks(dict):
    """The `stacks` dict of trace_skew(aggregate=True): three lists of S per-surface tensors, plus
    `q_sum` = sum over all rays of (sum theta + sum theta' + sum z_RELU) with NaN -> 0, fused into the
    trace kernel and differentiable through its backward kernel (the lists themselves are values only)."""

    def __init__(self, stk, moments, n_lens=1):
        super().__init__()
        if stk is not None:        # aggregate='sum': the fused sums only, no per-surface tensors (132 B per ray at 11 rows)
            for j, key in enumerate(('z_RELU', 'theta_norm', 'theta_prime_norm')):
                self[key] = list(torch.unbind(stk[j], dim=0))
        self._moments, self._n_lens = moments, n_lens

    @property
    def q_sum(self):
        return self._moments[:, 8].sum()

    @property
    def q_per_lens(self):                    # lens batch: one penalty sum per lens
        return self._moments[:, 8].view(self._n_lens, -1).sum(dim=1)


def penalty_sum(stacks, n_sequence: int):
    """sumQ of the reference's compute_loss_out (optics_simulator_lite.py:441-448):
    Q = (sum_k theta + sum_k theta' + sum_k z_RELU) / n_sequence per ray, NaN -> 0, summed over rays."""
    if isinstance(stacks, PenaltyStacks):
        return (stacks.q_sum / n_sequence).to(torch.float32)
    q = (torch.stack(stacks['theta_norm'], 0).sum(0) + torch.stack(stacks['theta_prime_norm'], 0).sum(0)
         + torch.stack(stacks['z_RELU'], 0).sum(0)) / n_sequence
    return torch.where(torch.isnan(q), torch.zeros_like(q), q).sum()


def unsupervised_loss(rt_outputs, n_sequence: int, penalty_rate: float):
    """loss_dict of the reference's RaytracedOptics.compute_loss_out (optics_simulator_lite.py:430-450):
    {'loss_unsup': rms + penalty_rate * sumQ, 'rms': rms, 'penalty': sumQ} from the 7-tuple that
    trace_rays(..., aggregate=True) returns."""
    x, y, *_, ray_ok, _ray_backward, stacks = rt_outputs
    fused = _fused_unsup_loss(y, ray_ok, stacks, n_sequence, penalty_rate, 1)
    if fused is not None:
        return fused
    rms = compute_rms2d(x, y, ray_ok)
    pen = penalty_sum(stacks, n_sequence)
    return {'loss_unsup': rms + penalty_rate * pen, 'rms': rms, 'penalty': pen}


def _fused_unsup_loss(y, ray_ok, stacks, n_sequence, penalty_rate, n_lens):
    """The loss_dict in ONE launch (tl_unsup_loss, C++ host chain) when everything it needs is the fused moments of an
    aggregate trace of fp32 tensors on the GPU; None otherwise (the callers compose it from tensor ops: same values)."""
    tag = getattr(y, "_tl_spot", None)
    if (not isinstance(stacks, PenaltyStacks) or tag is None or tag[1] is not ray_ok or tag[2] != y._version
            or stacks._moments is not tag[0] or y.dtype != torch.float32 or not isinstance(penalty_rate, (int, float))):
        return None
    moments = tag[0]
    if n_lens == 1 and y.shape[0] > 1:
        return None                     # compute_rms2d reads sample 0 of a batch, the penalty sums all of it: not this kernel
    out = ops.unsup_loss(moments, tag[3], n_lens, n_sequence, penalty_rate)
    if out is None:
        return None
    return {'loss_unsup': out[0], 'rms': out[1], 'penalty': out[2]}


def unsupervised_loss_batch(rt_outputs, n_sequence, penalty_rate: float):
    """Extension: the loss_dict of compute_loss_out for EVERY lens of a batch, each entry [B] -- what the reference's
    caller computes one lens at a time in a Python loop over its minibatch (optical_loss.py:96-110), here from ONE
    batched trace.  `n_sequence`: rows of the sequence string per lens (an int, or a [B] tensor for padded batches:
    compute_loss_out divides by len(self._sequence[0]))."""
    x, y, *_, ray_ok, _ray_backward, stacks = rt_outputs
    fused = _fused_unsup_loss(y, ray_ok, stacks, n_sequence, penalty_rate, y.shape[0])
    if fused is not None:
        return fused
    rms = compute_rms2d_batch(x, y, ray_ok)
    if isinstance(stacks, PenaltyStacks):
        q = stacks.q_per_lens
    else:
        q = (torch.stack(stacks['theta_norm'], 0).sum(0) + torch.stack(stacks['theta_prime_norm'], 0).sum(0)
             + torch.stack(stacks['z_RELU'], 0).sum(0))
        q = torch.where(torch.isnan(q), torch.zeros_like(q), q).sum(dim=(1, 2, 3))
    # (a Python number divides without a host-to-device copy: the step can sit inside a captured HIP graph)
    n_seq = n_sequence if isinstance(n_sequence, (int, float)) else torch.as_tensor(n_sequence, device=q.device, dtype=q.dtype)
    pen = (q / n_seq).to(torch.float32)
    return {'loss_unsup': rms + penalty_rate * pen, 'rms': rms, 'penalty': pen}


def rms_from_moments(moments: torch.Tensor, n_per_field: int) -> torch.Tensor:
    """compute_rms2d written on the per-field moments (fp64): with m = M0/n,
    rms = mean_f sqrt((M2 - 2 m M1 + m^2 M3) / n).  Failed rays count as y = 0 in the
    centroid and in the denominator, exactly like the reference (quirk B8)."""
    m = moments[:, 0] / n_per_field
    var = (moments[:, 2] - 2 * m * moments[:, 1] + m * m * moments[:, 3]) / n_per_field
    return torch.sqrt(var).mean()


def compute_rms2d(x, y, ray_ok, group=None, n_per_field: Optional[int] = None):
    """Mean over fields of the y-RMS spot radius of lens 0 (ray_tracing_lite.py:678-702).

    `group`: a torch.distributed process group over which the pupil dimension is sharded; the
    per-field moments are summed across it (one tiny all-reduce) before the closed form.
    `n_per_field`: P*W of the WHOLE (unsharded) pupil; defaults to this shard's P*W times the
    group size.
    """
    tag = getattr(y, "_tl_spot", None)
    if tag is not None and tag[1] is ray_ok and tag[2] == y._version:
        moments, n_local = tag[0], tag[3]
        if group is None and len(tag) > 5 and tag[5] is not None and n_per_field in (None, n_local) and y.dtype == torch.float32:
            return tag[5] if y.shape[0] == 1 else tag[5][0]        # computed by the trace node itself (one launch, one node)
        if y.shape[0] > 1:                       # the reference reads sample 0 only (:695,699)
            moments = moments[: y.shape[1]]
    else:
        if y.shape[0] > 1:
            x, y, ray_ok = (None if x is None else x[:1]), y[:1], ray_ok[:1]
        moments = ops.SpotMomentsFunction.apply(x, y, ray_ok)
        n_local = y.shape[2] * y.shape[3]
    if group is not None:
        from . import dist as tl_dist
        moments = tl_dist.all_reduce_sum(moments, group)
        if n_per_field is None:
            n_per_field = n_local * torch.distributed.get_world_size(group)
    if moments.is_cuda and y.dtype != torch.float64:
        return ops.spot_rms(moments, n_per_field or n_local).to(y.dtype)
    return rms_from_moments(moments, n_per_field or n_local).to(y.dtype)        # fp64 callers: the closed form in fp64


def compute_rms2d_batch(x, y, ray_ok):
    """Extension: compute_rms2d of EVERY lens of a batch, [B] (the reference's compute_rms2d reads sample 0 only and
    its callers loop over lenses, optical_loss.py:96-110).  Same closed form per lens on the moments fused into the
    one batched trace launch; differentiable through the one batched backward launch."""
    B, F = y.shape[0], y.shape[1]
    tag = getattr(y, "_tl_spot", None)
    if tag is not None and tag[1] is ray_ok and tag[2] == y._version:
        moments, n_local = tag[0], tag[3]
        if len(tag) > 5 and tag[5] is not None and y.dtype == torch.float32:
            return tag[5].reshape(B)                               # computed by the trace node itself
    else:
        fold = lambda a: None if a is None else a.reshape(1, B * F, a.shape[2], a.shape[3])      # noqa: E731
        moments = ops.SpotMomentsFunction.apply(fold(x), fold(y), fold(ray_ok))
        n_local = y.shape[2] * y.shape[3]
    if moments.is_cuda and y.dtype != torch.float64:
        return ops.spot_rms(moments, n_local, B).reshape(B).to(y.dtype)
    m = moments.view(B, F, -1)
    mean = m[..., 0] / n_local
    var = (m[..., 2] - 2 * mean * m[..., 1] + mean * mean * m[..., 3]) / n_local
    return torch.sqrt(var).mean(dim=1).to(y.dtype)


def compute_rms_spot_xy(x, y, ray_ok, group=None, n_per_field: Optional[int] = None):
    """Extension (not in the reference, whose compute_rms2d ignores x): mean over fields of the 2-D RMS
    spot radius sqrt(<(x - x_c)^2 + (y - y_c)^2>) with the same conventions as compute_rms2d (centroid over
    all rays, failed rays at the origin, denominator P*W).  Uses the x- and y-moments fused into the trace
    kernel; differentiable through the same backward kernel."""
    tag = getattr(y, "_tl_spot", None)
    if tag is not None and tag[1] is ray_ok and tag[2] == y._version and tag[4]:      # traced with x_moments=True
        moments, n_local = tag[0], tag[3]
    else:
        moments = ops.SpotMomentsFunction.apply(x, y, ray_ok)
        n_local = y.shape[2] * y.shape[3]
    if group is not None:
        from . import dist as tl_dist
        moments = tl_dist.all_reduce_sum(moments, group)
        if n_per_field is None:
            n_per_field = n_local * torch.distributed.get_world_size(group)
    n = n_per_field or n_local
    my, mx = moments[:, 0] / n, moments[:, 4] / n
    var = ((moments[:, 2] - 2 * my * moments[:, 1] + my * my * moments[:, 3])
           + (moments[:, 6] - 2 * mx * moments[:, 5] + mx * mx * moments[:, 3])) / n
    return torch.sqrt(var).mean().to(y.dtype)


# ---------------------------------------------------------------------------- RayTracer
class RayTracer:
    """Builds the ray fan for a `Specs`/`Lens` pair and traces it (ray_tracing_lite.py:26-208)."""

    def __init__(self, mode='skew_random', n_rays=(8, 8), rel_fields=(0., 0.707, 1.), vig_fn=None,
                 double_precision=False, wavelengths=(656.3, 587.6, 486.1), n_ray_aiming_iter=0,
                 ray_aiming_mode='real', allow_backward_rays=True, default_device='cuda', arith=None):
        self.mode = mode
        self.default_device = default_device
        dev = default_device
        two = lambda fn: (lambda tensor: fn(tensor, *n_rays, device=dev))      # noqa: E731  (n_r, n_theta)-style
        one = lambda fn: (lambda tensor: fn(tensor, n_rays, device=dev))       # noqa: E731  count-style
        spans = {
            'skew_random': lambda tensor: circle_pseudo_random(tensor, *n_rays),
            'circular': lambda tensor: circle(tensor, *n_rays, dev),
            'tee': lambda tensor: tee(tensor, dev),
            'skew_uniform_half_equidistant': two(skew_uniform_half_equidistant),
            'skew_uniform_half_jittered': two(skew_uniform_half_jittered),
            'skew_inner_square_half': two(skew_inner_square_half),
            'skew_outer_edge_uniform': one(circle_outer_edge_uniform),
            'meridional_uniform': one(meridional_uniform),
            'sagittal_uniform': one(sagittal_uniform),
            'chief': one(chief),
        }
        if mode not in spans:
            raise ValueError(f"Ray tracing mode must be one of {_MODES}")        # the reference `assert`s a ValueError (B6)
        if mode in ('skew_random', 'circular', 'skew_uniform_half_equidistant', 'skew_uniform_half_jittered',
                    'skew_inner_square_half'):
            assert len(n_rays) == 2
        self.pupil_span = spans[mode]
        self.n_rays = n_rays
        self.rel_fields = rel_fields
        self.vig_fn = vig_fn
        self.n_ray_aiming_iter = n_ray_aiming_iter
        self.ray_aiming_mode = ray_aiming_mode
        self.allow_backward_rays = allow_backward_rays
        self.wavelengths = [_LINES.get(w, w) for w in wavelengths]
        self.double_precision = double_precision
        self.arith = arith      # None -> ops.get_default_mode()

    # -- host-side assembly of the kernel arguments (ray_tracing_lite.py:86-124) -------------
    def assemble(self, specs, lens, xy=None, up_to_stop=False, use_vig=True):
        """Everything `trace_skew` needs, as a dict: x, y, z, cx, cy, c, t, mu, mask."""
        dev = self.default_device
        if self.double_precision and lens.c.dtype != torch.float64:
            # ray_tracing_lite.py:82-84 (which crashes in the reference: its Specs / Lens have no .double()): everything
            # downstream in fp64 -- dispersion, ABCD chain, fan, and the generic fp64 trace kernels (tl_trace_*_f64)
            specs, lens = specs.double(), lens.double()
        from .lens_modeling import _memoised
        n_rows = lens.get_refractive_indices(self.wavelengths)            # [1, S, W]

        def with_object_space():
            m = torch.cat((torch.ones_like(n_rows[:, :1, :]), n_rows), dim=1).transpose(1, 2)
            return m.reshape(m.shape[0], 1, 1, m.shape[1], -1)            # [1,1,1,W,S+1]
        # (constant glasses: the dispersion result is the same tensor every step, and so is everything derived from it)
        n = _memoised("n_index", (), (n_rows,), (n_rows,), with_object_space)
        strict = (self.arith or ops.get_default_mode()) == "strict"
        # the rows in front of the stop are cut out once: the pupil position and the ray aiming both start from them
        # (each Lens built on the way is ~10 tiny host-side view ops; this chain is what bounds a small minibatch)
        front = lens.up_to_stop()
        z1 = compute_pupil_position(lens, self.arith, front=front)
        z = z1.reshape(-1, 1, 1, 1)
        xy_abs = None
        xp_rel, yp_rel = self.pupil_span(z) if xy is None else xy
        if use_vig and self.vig_fn is not None and self.mode != 'chief':
            yp_rel, xp_rel = self._vignette(specs, yp_rel.to(dev), xp_rel.to(dev))
        if self.n_ray_aiming_iter > 0 and not up_to_stop:
            aim = self.ray_aiming(specs, lens, use_vig, front=front, z=z1.detach())
            fan = getattr(aim, "fan", None)        # the aiming kernel's one-launch form of the three steps below
            xy_abs = fan(xp_rel, yp_rel, specs.epd) if fan is not None else None
            if xy_abs is None:
                xp_rel, yp_rel = (torch.clamp(v, -2, 2).to(dev).detach() for v in aim(xp_rel, yp_rel))
        from .lens_modeling import const_tensor
        fields = const_tensor(list(self.rel_fields), torch.float32, dev)
        # strict: the correctly rounded fp32 sine (evaluated in fp64, rounded once) -- the reference's value on every CPU
        # whose libm rounds correctly there, and the fixtures' bit for bit; fast: the device's fp32 sine
        def sines():
            ang = (specs.hfov[:, None] * fields[None, :])[..., None, None]
            return torch.sin(ang.double()).to(ang.dtype) if (strict and ang.is_cuda and ang.dtype == torch.float32) else torch.sin(ang)
        cy = _memoised("cy", (strict,), (specs.hfov, fields), (specs.hfov,), sines)
        cx = const_tensor([0.0], torch.float32, dev, (1, 1, 1, 1))
        x_abs, y_abs = xy_abs if xy_abs is not None else (scale_to_epd(xp_rel.to(dev), specs.epd), scale_to_epd(yp_rel.to(dev), specs.epd))
        out = dict(
            x=x_abs, y=y_abs, z=z, cx=cx, cy=cy,
            c=lens.c.reshape(lens.c.shape[0], 1, 1, 1, -1), t=lens.t.reshape(lens.t.shape[0], 1, 1, 1, -1),
            mu=_memoised("mu", (), (n,), (n,), lambda: n[..., :-1] / n[..., 1:]),
            mask=lens.structure.mask_torch.reshape(lens.c.shape[0], 1, 1, 1, -1))
        if getattr(lens, "kappa", None) is not None:        # aspheric extension of Lens (not in the reference)
            out.update(kappa=lens.kappa, poly=lens.poly, n_index=n)
        return out

    def _vignette(self, specs, yp_rel, xp_rel):
        """Per-field vignetting of the relative pupil coordinates (ray_tracing.py:479-490 of the TF original):
        `vig_fn(fields [1,F], v [B]) -> [B,F]` interpolates the three vignetting factors over the field."""
        fields = torch.tensor(self.rel_fields, dtype=torch.float32, device=yp_rel.device)[None, :]
        up, down, vx = (self.vig_fn(fields, v) for v in (specs.vig_up, specs.vig_down, specs.vig_x))
        return apply_vignetting(yp_rel, up, down), apply_vignetting(xp_rel, vx, vx)

    def trace_rays(self, specs, lens, use_vig=True, aggregate=False, xy=None, up_to_stop=False, want_opd=False,
                   x_moments=False):
        a = self.assemble(specs, lens, xy=xy, up_to_stop=up_to_stop, use_vig=use_vig)
        extra = {}
        if "kappa" in a:
            extra.update(kappa=a["kappa"], poly=a["poly"])
        if want_opd:
            n = a.get("n_index")
            if n is None:
                n = lens.get_refractive_indices(self.wavelengths)
                n = torch.cat((torch.ones_like(n[:, :1, :]), n), dim=1).transpose(1, 2)
                n = n.reshape(n.shape[0], 1, 1, n.shape[1], -1)
            extra.update(n_index=n, want_opd=True)
        if x_moments:
            extra.update(x_moments=True)
        return trace_skew(a['x'], a['y'], a['z'], a['cx'], a['cy'], a['c'], a['t'], a['mu'], a['mask'],
                          aggregate, self.allow_backward_rays, mode=self.arith, **extra)

    def _ray_aiming_kernel(self, specs2stop, lens2stop, z=None):
        """ray_aiming as ONE kernel (tl_ray_aim): marginal ray, the three tee rays with their Jacobian, the Newton step
        and the affine pupil map, per (lens, field, wavelength) in fp64 registers -- instead of two eager traces, an
        autograd pass and ~60 tensor ops (1.1 ms of host time per call for a 256-lens minibatch, 0.1 ms now)."""
        import ctypes as C
        from . import _lib
        from .lens_modeling import const_tensor
        dev = lens2stop.c.device
        B, K = lens2stop.c.shape
        F, W = len(self.rel_fields), len(self.wavelengths)
        with torch.no_grad():              # (lens2stop may still be attached to the caller's graph: values only here)
            n = _dense(lens2stop.get_refractive_indices(self.wavelengths).detach())            # [B,K,W]
            n_d = _dense(lens2stop.get_refractive_indices([_LINES['d']]).detach())             # [B,K,1]
            # (z: the pupil position of these rows when the caller has it -- up_to_stop of the cut lens is the cut lens)
            z = _dense((compute_pupil_position(lens2stop, self.arith) if z is None else z).detach())
        c, t = _dense(lens2stop.c.detach()), _dense(lens2stop.t.detach())
        mask = _dense(lens2stop.structure.mask_torch.view(torch.uint8))
        kap = pol = kind = None
        if getattr(lens2stop, "kappa", None) is not None:
            kap, pol = _dense(lens2stop.kappa.detach().float()), _dense(lens2stop.poly.detach().float())
            kind = ((kap != 0) | (pol != 0).any(dim=-1)).to(torch.uint8)
        fields = const_tensor(list(self.rel_fields), torch.float32, dev)
        hfov, epd = _dense(specs2stop.hfov.detach().float()), _dense(specs2stop.epd.detach().float())
        out = torch.empty((3, B, F, 1, W), dtype=torch.float32, device=dev)
        with ops._on_device(dev):
            rc = _lib.lib().tl_ray_aim(dev.index, B, F, W, K, _lib.ptr(c), _lib.ptr(t), _lib.ptr(n), _lib.ptr(n_d), _lib.ptr(mask),
                                       _lib.ptr(kap), _lib.ptr(pol), _lib.ptr(kind), _lib.ptr(z), _lib.ptr(hfov), _lib.ptr(fields),
                                       _lib.ptr(epd), 1 if self.allow_backward_rays else 0, _lib.ptr(out[0]), _lib.ptr(out[1]),
                                       _lib.ptr(out[2]), ops._stream_ptr(dev))
        _lib.check(rc, "tl_ray_aim")
        x_scale, y_scale, y_offset = out[0], out[1], out[2]

        def remap(xp_rel, yp_rel):
            return xp_rel * x_scale, yp_rel * y_scale + y_offset

        def fan(xp_rel, yp_rel, epd_full):
            """clamp(remap(.), -2, 2) scaled to the pupil, [B,F,P,W] (memory [B,F,W,P]), in ONE launch (tl_aim_fan) -- or
            None when the pupil grid is not one shared fp32 [1,1,P,1] fan (the caller then composes it from tensor ops)."""
            P = xp_rel.shape[2] if xp_rel.dim() == 4 else 0
            if (P == 0 or tuple(xp_rel.shape) != (1, 1, P, 1) or tuple(yp_rel.shape) != (1, 1, P, 1) or B * F * W > 65535
                    or not (xp_rel.is_cuda and yp_rel.is_cuda and xp_rel.dtype == yp_rel.dtype == torch.float32)
                    or epd_full.requires_grad or xp_rel.requires_grad or yp_rel.requires_grad):
                return None
            xp, yp, e = _dense(xp_rel.detach()), _dense(yp_rel.detach()), _dense(epd_full.detach().float())
            xy = torch.empty((2, B, F, W, P), dtype=torch.float32, device=dev)
            with ops._on_device(dev):
                rc2 = _lib.lib().tl_aim_fan(dev.index, B, F, W, P, _lib.ptr(xp), _lib.ptr(yp), _lib.ptr(x_scale), _lib.ptr(y_scale),
                                            _lib.ptr(y_offset), _lib.ptr(e), _lib.ptr(xy[0]), _lib.ptr(xy[1]), ops._stream_ptr(dev))
            _lib.check(rc2, "tl_aim_fan")
            return xy[0].permute(0, 1, 3, 2), xy[1].permute(0, 1, 3, 2)
        remap.fan = fan
        return remap

    # -- ray aiming (ray_tracing_lite.py:129-208) ---------------------------------------------
    def ray_aiming(self, specs, lens, use_vig, front=None, z=None):
        """One Newton step per iteration on the pupil coordinates of three 'tee' rays so that
        they land on the stop where an ideal pupil would put them; returns the affine pupil
        remap.  The Jacobian diagonal comes from the backward kernel's per-ray input grads.
        `front`, `z`: lens.up_to_stop() and its pupil position [B] if the caller has them already (`lens` and `front`
        may then still be attached to a graph: nothing here differentiates through them)."""
        if (lens.structure.stop_idx == 0).all():
            return lambda xp_rel, yp_rel: (xp_rel, yp_rel)
        if self.n_ray_aiming_iter > 1:
            raise NotImplementedError("n_ray_aiming_iter >= 2 fails in the reference as well (Appendix B4)")
        specs2stop, lens2stop = specs.up_to_stop(), (lens.up_to_stop() if front is None else front)
        if (self.ray_aiming_mode == 'real' and lens2stop.c.is_cuda and lens2stop.c.dtype == torch.float32
                and not (use_vig and self.vig_fn is not None) and _AIM_KERNEL):
            return self._ray_aiming_kernel(specs2stop, lens2stop, z)
        if front is not None:
            lens2stop = lens2stop.detach()
        if self.ray_aiming_mode == 'paraxial':
            rs = (compute_magnification(lens2stop) * specs2stop.epd / 2).reshape(-1, 1, 1, 1)
        elif self.ray_aiming_mode == 'real':
            rs = compute_pupil_radius(specs2stop, lens2stop, default_device=self.default_device).reshape(-1, 1, 1, 1)
        else:
            raise ValueError("ray_aiming_mode must be 'real' or 'paraxial'")

        xt0, yt0 = tee(None, self.default_device)
        shape = (len(lens), len(self.rel_fields), xt0.shape[2], len(self.wavelengths))
        xt_ref, yt_ref = xt0.expand(shape).clone(), yt0.expand(shape).clone()
        if use_vig and self.vig_fn is not None:
            yt_ref, xt_ref = self._vignette(specs, yt_ref, xt_ref)
        with torch.enable_grad():
            xt = xt_ref.clone().requires_grad_(True)
            yt = yt_ref.clone().requires_grad_(True)
            xs, ys, *_ = self.trace_rays(specs2stop, lens2stop, up_to_stop=True, use_vig=False, xy=(xt, yt))
            xs_rel, ys_rel = xs / rs, ys / rs
            # the reference back-propagates ones through xs_rel and then ys_rel into the same .grad
            jx, jy = torch.autograd.grad(xs_rel.sum() + ys_rel.sum(), (xt, yt))
        xs_rel, ys_rel, xt, yt = xs_rel.detach(), ys_rel.detach(), xt.detach(), yt.detach()
        dx = -(xs_rel - xt_ref) / jx
        dy = -(ys_rel - yt_ref) / jy
        dx = torch.where(torch.isfinite(dx), dx, torch.zeros_like(dx))
        dy = torch.where(torch.isfinite(dy), dy, torch.zeros_like(dy))
        dx_s = dx[..., -1:, :]
        dy_l, dy_u = dy[..., 0:1, :], dy[..., 1:2, :]
        x_s = xt[..., -1:, :]
        y_l, y_u = yt[..., 0:1, :], yt[..., 1:2, :]
        y_scale = (y_u + dy_u - (y_l + dy_l)) / (y_u - y_l)
        y_offset = (y_l * dy_u - y_u * dy_l) / (y_l - y_u)

        def remap(xp_rel, yp_rel):
            return xp_rel * (x_s + dx_s) / x_s, yp_rel * y_scale + y_offset
        return remap
