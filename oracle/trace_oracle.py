"""
TEST INFRASTRUCTURE ONLY -- CPU oracle for the sequential ray-trace hot path.

This module is a CPU restatement (eager PyTorch, autograd-recorded backward) of the
algorithm in the reference `torchlens/ray_tracing_lite.py`.  It exists so that the HIP
kernels in `torchoptics_amd/csrc/` can be checked value-for-value; it is NOT part of the
product.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it.  The product (`torchoptics_amd`) never imports it and fails
loudly when its HIP library is missing.

Pinning: `tests/golden/make_golden.py` ran the reference itself (imported from
/root/reference in the build container) and committed its outputs as fixtures under
`tests/golden/`; `tests/test_oracle_golden.py` checks this file against them (forward
bit-exact in fp32, gradients equal).  The spherical path is therefore pinned by the
reference.  The aspheric / Newton / OPD path (`trace_skew_general`) is an extension the
reference does not have: PARITY UNPINNED by the reference for that part -- it is the
definition the kernels are tested against.

Reference map (file = torchlens/ray_tracing_lite.py):
  sphere_hit        <- find_marching_distance_spherical  :525-545
  advance           <- update_ray_coordinates            :514-522
  retire_dead       <- reset_bad_rays (normalize=False)  :574-591
  refract_sphere    <- apply_snell_spherical             :548-571
  trace_skew        <- trace_skew                        :594-675
  compute_rms2d     <- compute_rms2d                     :678-702
The fp32 operation ORDER follows SURVEY.md Appendix A so that the forward is bit-exact
with the reference on CPU.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

EPS = 1e-6          # ray_tracing_lite.py:530,552
ACOS_EPS = 1e-7     # ray_tracing_lite.py:644


# ----------------------------------------------------------------------------------
# Square root.  torch.sqrt on CPU tensors goes through MKL VML for large fp32 tensors and is
# NOT correctly rounded there (measured in the build container: 0.7 % of uniform(0.5,1) inputs
# are 1 ulp off, while short tensors take another code path and are exact), so the reference's
# own fp32 output depends on the tensor size and cannot be reproduced bit for bit by IEEE-754
# arithmetic.  The oracle therefore has two settings:
#   ieee_sqrt=False : torch.sqrt  -> bit-exact with the reference fixtures (pins the oracle)
#   ieee_sqrt=True  : correctly rounded sqrt (numpy) -> what the strict HIP kernels must match
#                     bit for bit; it differs from the reference by <= a few ulp on a few % of rays
# ----------------------------------------------------------------------------------
class _IeeeSqrt(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v):
        import numpy as np
        out = torch.from_numpy(np.sqrt(v.detach().cpu().contiguous().numpy())).to(v.device)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        return g / (2 * out)


_IEEE = False


def _sqrt(v):
    return _IeeeSqrt.apply(v) if (_IEEE and v.dtype == torch.float32) else torch.sqrt(v)


@dataclass
class RayBundle:
    """Structure-of-arrays ray state in the local frame of the next surface vertex."""
    x: torch.Tensor
    y: torch.Tensor
    z: torch.Tensor
    cx: torch.Tensor
    cy: torch.Tensor
    cz: torch.Tensor
    ok: torch.Tensor      # sticky "still alive" flag
    back: torch.Tensor    # "travelled backwards somewhere" flag


def sphere_hit(c: torch.Tensor, r: RayBundle):
    """Closed-form ray/sphere marching distance (ref :525-545).

    Returns (miss, d, cos_i, cos2_i).  cos_i is sqrt of a value forced to 1 where the
    ray misses, which cuts the gradient for missed rays exactly as the reference does.
    """
    e = -(r.x * r.cx + r.y * r.cy + r.z * r.cz)
    mz = r.z + e * r.cz
    m2 = r.x ** 2 + r.y ** 2 + r.z ** 2 - e ** 2
    tmp = c * m2 - 2 * mz
    cos2_i = r.cz ** 2 - c * tmp
    miss = cos2_i - EPS < 0
    cos_i = _sqrt(torch.where(miss, torch.ones_like(cos2_i), cos2_i))
    d = e + tmp / (r.cz + cos_i)
    return miss, d, cos_i, cos2_i


def advance(r: RayBundle, d: torch.Tensor) -> torch.Tensor:
    """Move the bundle along its direction by d (ref :514-522); returns delta-z."""
    dz = d * r.cz
    r.x = r.x + d * r.cx
    r.y = r.y + d * r.cy
    r.z = r.z + dz
    return dz


def retire_dead(r: RayBundle) -> None:
    """Dead rays are parked at the vertex pointing along +z (ref :574-591)."""
    zero = torch.zeros((), dtype=r.x.dtype, device=r.x.device)
    one = torch.ones((), dtype=r.x.dtype, device=r.x.device)
    r.x = torch.where(r.ok, r.x, zero)
    r.y = torch.where(r.ok, r.y, zero)
    r.z = torch.where(r.ok, r.z, zero)
    r.cx = torch.where(r.ok, r.cx, zero)
    r.cy = torch.where(r.ok, r.cy, zero)
    r.cz = torch.where(r.ok, r.cz, one)


def refract_sphere(c: torch.Tensor, mu: torch.Tensor, r: RayBundle, cos_i: torch.Tensor):
    """Snell refraction at a spherical interface (ref :548-571).

    Updates r.cx, r.cy, r.cz; returns (fail, cos2_t).
    """
    cos2_t = 1 - mu ** 2 * (1 - cos_i ** 2)
    tir = cos2_t - EPS < 0
    cos_t = _sqrt(torch.where(tir, torch.ones_like(cos2_t), cos2_t))
    g = cos_t - mu * cos_i
    ncx = mu * r.cx - g * c * r.x
    ncy = mu * r.cy - g * c * r.y
    cz2 = 1 - (ncx ** 2 + ncy ** 2)
    fail = tir | (cz2 - EPS < 0)
    r.cx, r.cy = ncx, ncy
    r.cz = _sqrt(torch.where(fail, torch.ones_like(cz2), cz2))
    return fail, cos2_t


def _flag_backward(r: RayBundle, dz: torch.Tensor, live: torch.Tensor, allow: bool) -> None:
    """Backward-travel bookkeeping (ref :626-632, :665-670)."""
    hit = (dz < 0) & live
    if allow:
        r.back = r.back | hit
    else:
        r.ok = r.ok & ~hit


def trace_skew(x, y, z, cx, cy, c, t, mu, mask, aggregate: bool = False,
               allow_backward_rays: bool = True, ieee_sqrt: bool = False):
    """Sequential trace through S spherical surfaces to the image plane (ref :594-675).
    `ieee_sqrt`: see the note on _IeeeSqrt above (False = reference-exact).

    Shapes as in the reference: x,y [1|B,1|F,P,1|W]; z [B,1,1,1]; cx [1,1,1,1];
    cy [B,F,1,1]; c,t [B,1,1,1,S]; mu [B,1,1,W,S]; mask [B,1,1,1,S] (bool).
    Returns (x, y, cx, cy, ray_ok, ray_backward[, stacks]).
    """
    global _IEEE
    prev, _IEEE = _IEEE, bool(ieee_sqrt)
    try:
        return _trace_skew(x, y, z, cx, cy, c, t, mu, mask, aggregate, allow_backward_rays)
    finally:
        _IEEE = prev


def _push_stacks(stacks, r: RayBundle, cos2_i, cos2_t, n_wave: int) -> None:
    """Penalty-term raw material of one surface row (ref :641-657): z_RELU, theta/(pi/2), theta'/(pi/2)."""
    full = (*r.x.shape[:3], n_wave)
    z_relu = torch.where(r.z <= 0, torch.zeros_like(r.z), r.z)
    lo, hi = -1.0 + ACOS_EPS, 1.0 - ACOS_EPS
    # The reference takes sqrt(cos2) of EVERY ray and overwrites theta of the dead ones with 1
    # afterwards (in place, which only works when shapes already agree: SURVEY Appendix B5).
    # For a missed ray cos2 < 0, sqrt is NaN, and although the forward value is overwritten the
    # backward multiplies a zero gradient by NaN: the reference's penalty gradient is NaN as soon
    # as one ray misses.  Same forward values here, but the sqrt argument of dead rays is
    # replaced by 1 first, so masked rays contribute exactly zero gradient.
    ok_b = torch.broadcast_to(r.ok, full)
    one = torch.ones((), dtype=r.x.dtype, device=r.x.device)
    safe_i = torch.where(ok_b, torch.broadcast_to(cos2_i, full), one)
    safe_t = torch.where(ok_b, torch.broadcast_to(cos2_t, full), one)
    th_i = torch.acos(torch.clamp(torch.sqrt(safe_i), min=lo, max=hi)) / (1 / 2 * math.pi)
    th_t = torch.acos(torch.clamp(torch.sqrt(safe_t), min=lo, max=hi)) / (1 / 2 * math.pi)
    th_i = torch.where(ok_b, th_i, one)
    th_t = torch.where(ok_b, th_t, one)
    stacks['z_RELU'].append(torch.broadcast_to(z_relu, full))
    stacks['theta_norm'].append(th_i)
    stacks['theta_prime_norm'].append(th_t)


def _trace_skew(x, y, z, cx, cy, c, t, mu, mask, aggregate, allow_backward_rays):
    n_surf = t.shape[-1]
    cs, ts, mus, masks = (torch.unbind(a, dim=-1) for a in (c, t, mu, mask))

    r = RayBundle(x, y, z, cx, cy, _sqrt(1 - cx ** 2 - cy ** 2),
                  torch.ones_like(y, dtype=torch.bool), torch.zeros_like(y, dtype=torch.bool))
    stacks: Dict[str, List[torch.Tensor]] = {'z_RELU': [], 'theta_norm': [], 'theta_prime_norm': []}
    n_wave = mus[0].shape[-1]

    for k in range(n_surf):
        miss, d, cos_i, cos2_i = sphere_hit(cs[k], r)
        dz = advance(r, d)
        r.ok = r.ok & ~miss
        retire_dead(r)
        fail, cos2_t = refract_sphere(cs[k], mus[k], r, cos_i)
        if k > 0:
            _flag_backward(r, dz, r.ok & masks[k - 1], allow_backward_rays)
        r.ok = r.ok & ~fail
        retire_dead(r)
        r.z = r.z - ts[k]

        if aggregate:
            _push_stacks(stacks, r, cos2_i, cos2_t, n_wave)

    # transfer to the image plane (ref :659-663)
    dz = -r.z
    dist = dz / r.cz
    r.x = r.x + dist * r.cx
    r.y = r.y + dist * r.cy
    _flag_backward(r, dz, r.ok & masks[-1], allow_backward_rays)

    if aggregate:
        return r.x, r.y, r.cx, r.cy, r.ok, r.back, stacks
    return r.x, r.y, r.cx, r.cy, r.ok, r.back


# ----------------------------------------------------------------------------------
# EXTENSION (not in the reference; PARITY UNPINNED by the reference): aspheric surfaces with a
# Newton intersection, vector Snell refraction at the aspheric normal, and optical path length.
#
#   sag(rho) = c rho / (1 + sqrt(1 - (1+kappa) c^2 rho)) + a4 rho^2 + a6 rho^3 + a8 rho^4 + a10 rho^5,
#   rho = x^2 + y^2.
#
# Rows with kind == 0 go through sphere_hit / refract_sphere above, untouched, so an all-spherical
# call of trace_skew_general is the reference algorithm.  This file is the DEFINITION the HIP
# kernels are tested against for kind == 1 rows: forward by value, backward by this autograd graph
# evaluated in fp64 (Newton unrolled for a fixed number of iterations, i.e. exact to rounding).
# ----------------------------------------------------------------------------------
NEWTON_ITERS = 8        # oracle: always this many (no early exit); kernels: at most this many
NEWTON_TOL = 1e-6       # fp32 convergence: |F| <= NEWTON_TOL * (1 + |z_hit|)   [mm]


def _sag_terms(c, kappa, a, rho):
    """Returns (sag, dsag/drho, conic-domain-violated)."""
    q2 = 1 - (1 + kappa) * c * c * rho
    bad = q2 - EPS < 0
    q = _sqrt(torch.where(bad, torch.ones_like(q2), q2))
    sag = c * rho / (1 + q) + rho * rho * (a[0] + rho * (a[1] + rho * (a[2] + rho * a[3])))
    dsag = c / (2 * q) + rho * (2 * a[0] + rho * (3 * a[1] + rho * (4 * a[2] + rho * (5 * a[3]))))
    return sag, dsag, bad


def conic_hit(c, kappa, r: RayBundle):
    """Closed-form intersection of the ray with the CONIC part of the surface: the quadric
    c (x^2 + y^2 + K z^2) - 2 z = 0, K = 1 + kappa (the sag formula solved for the surface), along the ray
    A s^2 + 2 B s + C = 0 with A = c (d.d)_K, B = c (r.d)_K - cz, C = c (r.r)_K - 2 z; the root nearer the vertex plane in
    its cancellation-free form s = C / (-B + sqrt(B^2 - A C)).  For kappa = 0 this is sphere_hit's distance (the same
    quadratic written differently).  Returns (miss, s): miss = the ray does not meet the quadric."""
    K = 1 + kappa
    e = -((r.x * r.cx + r.y * r.cy) + K * (r.z * r.cz))
    dd = (r.cx * r.cx + r.cy * r.cy) + K * (r.cz * r.cz)
    rr = (r.x * r.x + r.y * r.y) + K * (r.z * r.z)
    bq = c * e + r.cz
    cq = c * rr - 2 * r.z
    disc = bq * bq - (c * dd) * cq
    miss = disc - EPS < 0
    s = cq / (bq + _sqrt(torch.where(miss, torch.ones_like(disc), disc)))
    return miss, s


def asphere_hit(c, kappa, a, r: RayBundle):
    """Newton iteration on the ray parameter s from the closed-form hit on the conic part of the surface (round 3; the
    first definition started from the base SPHERE and called a ray that misses that sphere a miss even where it meets
    the conic -- a paraboloid or hyperboloid reaches further out than its vertex sphere).

    Returns (miss, s, X, Y, dsag, rho): hit point (X, Y), d sag / d rho there.
    """
    miss0, d0 = conic_hit(c, kappa, r)
    s = torch.where(miss0, torch.zeros_like(d0), d0)
    tol = NEWTON_TOL if s.dtype == torch.float32 else 1e-13
    for _ in range(NEWTON_ITERS):
        X, Y = r.x + s * r.cx, r.y + s * r.cy
        rho = X * X + Y * Y
        sag, dsag, _ = _sag_terms(c, kappa, a, rho)
        F = (r.z + s * r.cz) - sag
        Fp = r.cz - dsag * (2 * (X * r.cx + Y * r.cy))
        s = s - F / Fp
    X, Y = r.x + s * r.cx, r.y + s * r.cy
    rho = X * X + Y * Y
    sag, dsag, bad = _sag_terms(c, kappa, a, rho)
    zhit = r.z + s * r.cz
    conv = torch.abs(zhit - sag) <= tol * (1 + torch.abs(zhit))
    return miss0 | bad | ~conv, s, X, Y, dsag, rho


def refract_general(mu, r: RayBundle, dsag, rho):
    """Vector Snell at the aspheric normal n = (-m X, -m Y, 1)/sqrt(1 + m^2 rho), m = 2 dsag/drho.
    r.x, r.y are already the hit point.  Updates r.cx, r.cy, r.cz; returns (fail, cos2_i, cos2_t)."""
    m = 2 * dsag
    inv_n = 1 / _sqrt(1 + m * m * rho)
    nx, ny, nz = -(m * r.x) * inv_n, -(m * r.y) * inv_n, inv_n
    cos_i = (r.cx * nx + r.cy * ny) + r.cz * nz
    cos2_t = 1 - mu ** 2 * (1 - cos_i ** 2)
    tir = cos2_t - EPS < 0
    cos_t = _sqrt(torch.where(tir, torch.ones_like(cos2_t), cos2_t))
    g = cos_t - mu * cos_i
    ncx = mu * r.cx + g * nx
    ncy = mu * r.cy + g * ny
    cz2 = 1 - (ncx ** 2 + ncy ** 2)
    fail = tir | (cz2 - EPS < 0)
    r.cx, r.cy = ncx, ncy
    r.cz = _sqrt(torch.where(fail, torch.ones_like(cz2), cz2))
    return fail, cos_i * cos_i, cos2_t


def trace_skew_general(x, y, z, cx, cy, c, t, mu, mask, kappa=None, poly=None, kind=None,
                       allow_backward_rays: bool = True, n_index=None, ieee_sqrt: bool = False,
                       aggregate: bool = False):
    """trace_skew with optional aspheric rows and optical path length.

    kappa [S], poly [S,4] (a4, a6, a8, a10), kind: sequence of S ints (0 sphere closed form,
    1 Newton asphere); n_index [1,1,1,W,S+1] refractive indices (index 0 = object space) for OPD.
    Returns (x, y, cx, cy, ray_ok, ray_backward, opd) with opd = sum_k n_k d_k + n_S dist_image
    (None when n_index is None).  Dead rays: opd = 0.
    aggregate=True appends the penalty-term `stacks` (as trace_skew does); at an aspheric row theta comes
    from the cosine between the ray and the aspheric normal.
    """
    global _IEEE
    prev, _IEEE = _IEEE, bool(ieee_sqrt)
    try:
        n_surf = t.shape[-1]
        cs, ts, mus, masks = (torch.unbind(a, dim=-1) for a in (c, t, mu, mask))
        kind = [0] * n_surf if kind is None else [int(k) for k in kind]
        ns = None if n_index is None else torch.unbind(n_index, dim=-1)
        r = RayBundle(x, y, z, cx, cy, _sqrt(1 - cx ** 2 - cy ** 2),
                      torch.ones_like(y, dtype=torch.bool), torch.zeros_like(y, dtype=torch.bool))
        opd = None if ns is None else torch.zeros_like(y)
        stacks: Dict[str, List[torch.Tensor]] = {'z_RELU': [], 'theta_norm': [], 'theta_prime_norm': []}
        for k in range(n_surf):
            if kind[k] == 0:
                miss, d, cos_i, cos2_i = sphere_hit(cs[k], r)
                dz = advance(r, d)
            else:
                miss, d, X, Y, dsag, rho = asphere_hit(cs[k], kappa[k], poly[k], r)
                dz = d * r.cz
                r.x, r.y, r.z = X, Y, r.z + dz
            r.ok = r.ok & ~miss
            retire_dead(r)
            if kind[k] == 0:
                fail, cos2_t = refract_sphere(cs[k], mus[k], r, cos_i)
            else:
                fail, cos2_i, cos2_t = refract_general(mus[k], r, dsag, rho)
            if k > 0:
                _flag_backward(r, dz, r.ok & masks[k - 1], allow_backward_rays)
            r.ok = r.ok & ~fail
            retire_dead(r)
            r.z = r.z - ts[k]
            if opd is not None:
                opd = torch.where(r.ok, opd + ns[k] * d, torch.zeros_like(opd + ns[k] * d))
            if aggregate:
                _push_stacks(stacks, r, cos2_i, cos2_t, mus[0].shape[-1])
        dz = -r.z
        dist = dz / r.cz
        r.x = r.x + dist * r.cx
        r.y = r.y + dist * r.cy
        if opd is not None:
            opd = torch.where(r.ok, opd + ns[-1] * dist, torch.zeros_like(opd + ns[-1] * dist))
        _flag_backward(r, dz, r.ok & masks[-1], allow_backward_rays)
        if aggregate:
            return r.x, r.y, r.cx, r.cy, r.ok, r.back, opd, stacks
        return r.x, r.y, r.cx, r.cy, r.ok, r.back, opd
    finally:
        _IEEE = prev


def compute_rms2d(x, y, ray_ok):
    """y-only RMS spot for sample 0, averaged over fields (ref :678-702).

    Per field: centroid = mean over wavelengths of the mean over ALL pupil points
    (failed rays sit at y=0 and are counted); numerator sums only live rays; the
    denominator is P*W regardless.  `x` is accepted and ignored, as in the reference.
    """
    n_f, n_p, n_w = y.shape[1], y.shape[2], y.shape[3]
    total = 0.
    for yf, okf in zip(torch.unbind(y[0], dim=0), torch.unbind(ray_ok[0], dim=0)):
        cen = 0.
        for w in range(n_w):
            cen = cen + torch.mean(yf[:, w])
        cen = cen / n_w
        total = total + torch.sqrt(torch.sum((yf[okf] - cen) ** 2) / (n_p * n_w))
    return total / n_f


# ----------------------------------------------------------------------------------
# Closed forms used by the sharded (multi-GPU) path; checked against compute_rms2d.
# ----------------------------------------------------------------------------------

def spot_moments(y: torch.Tensor, ray_ok: torch.Tensor) -> torch.Tensor:
    """Per-field fp64 moments [F,4] = (sum y, sum ok*y, sum ok*y^2, sum ok) of sample 0."""
    yd = y[0].double()
    okd = ray_ok[0].double()
    return torch.stack((yd.sum(dim=(1, 2)), (okd * yd).sum(dim=(1, 2)),
                        (okd * yd * yd).sum(dim=(1, 2)), okd.sum(dim=(1, 2))), dim=1)


def rms_from_moments(m: torch.Tensor, n_per_field: int) -> torch.Tensor:
    """compute_rms2d rewritten on the four moments; fp64 in, fp64 out."""
    mean = m[:, 0] / n_per_field
    var = (m[:, 2] - 2 * mean * m[:, 1] + mean * mean * m[:, 3]) / n_per_field
    return torch.sqrt(var).mean()


def penalty_from_stacks(stacks, n_sequence: int) -> torch.Tensor:
    """sum over rays of Q (reference optics_simulator_lite.py:441-448)."""
    q = (torch.stack(stacks['theta_norm'], dim=0).sum(dim=0)
         + torch.stack(stacks['theta_prime_norm'], dim=0).sum(dim=0)
         + torch.stack(stacks['z_RELU'], dim=0).sum(dim=0)) / n_sequence
    q = torch.where(torch.isnan(q), torch.zeros_like(q), q)
    return torch.sum(q)
