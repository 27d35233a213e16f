"""
CPU self-checks of the oracle's aspheric extension (oracle.trace_skew_general).  The reference
has no aspheres, so nothing here is pinned by it (PARITY UNPINNED): the checks are internal
consistency -- the all-spherical general path IS the reference path, Newton on a sphere finds
the closed-form hit, autograd agrees with finite differences in fp64, OPD obeys Fermat-style
sanity -- so that the definition the HIP kernels are tested against is itself trustworthy.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import trace_oracle as orc

IN = ("in_x", "in_y", "in_z", "in_cx", "in_cy", "in_c", "in_t", "in_mu")


def _case(dtype=torch.float32):
    g = load_golden("G4_tessar_32x32")
    return [torch.from_numpy(g[n]).to(dtype) for n in IN], torch.from_numpy(g["in_mask"])


def asphere_params(S, dtype=torch.float32):
    kap = torch.zeros(S, dtype=dtype)
    pol = torch.zeros(S, 4, dtype=dtype)
    kap[0], kap[5] = -0.8, 0.5
    pol[0, 0], pol[0, 1], pol[5, 0], pol[5, 2] = 2e-5, -3e-7, -4e-5, 1e-9
    kind = [0] * S
    kind[0] = kind[5] = 1
    return kap, pol, kind


def test_all_spherical_general_path_is_the_reference_path():
    ins, mask = _case()
    a = orc.trace_skew(*ins, mask)
    b = orc.trace_skew_general(*ins, mask)
    assert all(torch.equal(p, q) for p, q in zip(a, b[:6])) and b[6] is None


def test_newton_on_a_sphere_finds_the_closed_form_hit():
    ins, mask = _case(torch.float64)
    S = ins[5].shape[-1]
    kind = [1] * S
    kind[4] = 0                      # the flat stop stays closed-form
    a = orc.trace_skew(*ins, mask)
    b = orc.trace_skew_general(*ins, mask, torch.zeros(S, dtype=torch.float64), torch.zeros(S, 4, dtype=torch.float64), kind)
    assert torch.equal(a[4], b[4])
    assert (a[0] - b[0]).abs().max() < 1e-11 and (a[1] - b[1]).abs().max() < 1e-11


def test_autograd_of_aspheric_coefficients_matches_finite_differences():
    ins, mask = _case(torch.float64)
    S = ins[5].shape[-1]
    kap, pol, kind = asphere_params(S, torch.float64)
    kap.requires_grad_(True)
    pol.requires_grad_(True)
    c = ins[5].clone().requires_grad_(True)

    def loss():
        o = orc.trace_skew_general(ins[0], ins[1], ins[2], ins[3], ins[4], c, ins[6], ins[7], mask, kap, pol, kind)
        return orc.compute_rms2d(o[0], o[1], o[4])
    loss().backward()
    for tens, idx, h in ((kap, (0,), 1e-6), (kap, (5,), 1e-6), (pol, (0, 0), 1e-9), (pol, (0, 1), 1e-10),
                         (pol, (5, 2), 1e-13), (c, (0, 0, 0, 0, 0), 1e-7), (c, (0, 0, 0, 0, 5), 1e-7)):
        with torch.no_grad():
            base = tens[idx].item()
            tens[idx] = base + h
            lp = loss().item()
            tens[idx] = base - h
            lm_ = loss().item()
            tens[idx] = base
        fd = (lp - lm_) / (2 * h)
        assert abs(tens.grad[idx].item() - fd) <= 2e-5 * abs(fd) + 1e-12, (idx, tens.grad[idx].item(), fd)


def test_optical_path_length_is_consistent():
    ins, mask = _case(torch.float64)
    S = ins[5].shape[-1]
    mu = ins[7]                                            # [1,1,1,W,S] = n_before / n_after
    n = torch.ones(1, 1, 1, mu.shape[3], S + 1, dtype=torch.float64)
    for k in range(S):
        n[..., k + 1] = n[..., k] / mu[..., k]
    o = orc.trace_skew_general(*ins, mask, n_index=n)
    opd = o[6]
    assert opd.shape == o[0].shape and torch.isfinite(opd).all()
    # geometric length <= optical length (n >= 1), and the on-axis chief ray of field 0 runs straight
    # down the axis: its path from the pupil plane is  -z_pupil + sum_k n_k t_k (last t = image distance)
    axis = opd[0, 0, 0, :]                                  # pupil point 0 of the circular grid is r = 0
    want = -ins[2].reshape(()) * n[0, 0, 0, :, 0] + (n[0, 0, 0, :, 1:] * ins[6].reshape(1, S)).sum(dim=1)
    assert torch.allclose(axis, want, rtol=0, atol=1e-9)
    assert (opd[o[4]] > 0).all() and (opd[~o[4]] == 0).all()


def _n_index(mu, dtype):
    """[1,1,1,W,S+1] refractive indices with n_0 = 1 and n_k / n_{k+1} = mu_k."""
    S = mu.shape[-1]
    n = [torch.ones(1, 1, 1, mu.shape[3], dtype=dtype)]
    for k in range(S):
        n.append(n[-1] / mu[..., k])
    return torch.stack(n, dim=-1)


def test_general_aggregate_on_spherical_rows_is_the_reference_penalty():
    ins, mask = _case()
    a = orc.trace_skew(*ins, mask, aggregate=True)
    b = orc.trace_skew_general(*ins, mask, aggregate=True)
    for key in ("z_RELU", "theta_norm", "theta_prime_norm"):
        assert all(torch.equal(p, q) for p, q in zip(a[6][key], b[7][key])), key


def test_penalty_and_opd_gradients_on_aspheric_rows_match_finite_differences():
    """The two round-2 extensions of the definition: the penalty term with aspheric rows (theta from the cosine at
    the aspheric normal) and the gradient THROUGH the optical path length, incl. d/d n_index; fp64, central
    differences."""
    ins, mask = _case(torch.float64)
    S = ins[5].shape[-1]
    kap, pol, kind = asphere_params(S, torch.float64)
    n = _n_index(ins[7], torch.float64)
    kap.requires_grad_(True)
    c = ins[5].clone().requires_grad_(True)
    t = ins[6].clone().requires_grad_(True)
    n.requires_grad_(True)
    torch.manual_seed(3)
    w_opd = torch.rand(1, ins[4].shape[1], ins[0].shape[2], n.shape[3], dtype=torch.float64)

    def loss():
        o = orc.trace_skew_general(ins[0], ins[1], ins[2], ins[3], ins[4], c, t, ins[7], mask, kap, pol, kind,
                                   n_index=n, aggregate=True)
        return (o[6] * w_opd).sum() * 1e-3 + orc.penalty_from_stacks(o[7], S) * 1e-2
    loss().backward()
    for tens, idx, h in ((kap, (0,), 1e-6), (kap, (5,), 1e-6), (c, (0, 0, 0, 0, 0), 1e-7), (c, (0, 0, 0, 0, 3), 1e-7),
                         (t, (0, 0, 0, 0, 2), 1e-6), (n, (0, 0, 0, 1, 3), 1e-7), (n, (0, 0, 0, 0, S), 1e-7)):
        with torch.no_grad():
            base = tens[idx].item()
            tens[idx] = base + h
            lp = loss().item()
            tens[idx] = base - h
            lm_ = loss().item()
            tens[idx] = base
        fd = (lp - lm_) / (2 * h)
        assert abs(tens.grad[idx].item() - fd) <= 5e-5 * abs(fd) + 1e-9, (idx, tens.grad[idx].item(), fd)


# ---------------------------------------------------------------------------------------------------------------
# Analytic pins (round 3): geometry and Fermat's principle, independent of this oracle and of the kernels
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,tol_xy,tol_opd", [(torch.float64, 1e-11, 1e-11), (torch.float32, 1e-5, 1e-5)])
def test_stigmatic_conic_focuses_every_ray_and_equalises_the_optical_path(dtype, tol_xy, tol_opd):
    import analytic_asphere as an
    for pad in (0, 2):
        args, extra = an.stigmatic_conic(dtype, pad_rows=pad)
        o = orc.trace_skew_general(*args, extra["kappa"], extra["poly"], extra["surf_kind"], n_index=extra["n_index"],
                                   ieee_sqrt=(dtype == torch.float32))
        x, y, ok, opd = o[0], o[1], o[4], o[6]
        assert ok.all() and x.shape[2] > 3000
        assert x.abs().max().item() <= tol_xy and y.abs().max().item() <= tol_xy
        assert (opd - an.expected_opd(pad_rows=pad)).abs().max().item() <= tol_opd      # Fermat: one optical path for every ray
        # ... and a SPHERE of the same radius does neither (the case is not vacuous)
        o = orc.trace_skew_general(*args, torch.zeros_like(extra["kappa"]), extra["poly"], extra["surf_kind"], n_index=extra["n_index"])
        assert o[0].abs().max().item() > 0.05 and (o[6] - an.expected_opd(pad_rows=pad)).abs().max().item() > 1e-3


def test_spot_size_gradient_changes_sign_at_the_stigmatic_conic_constant():
    import analytic_asphere as an
    grads = []
    for dk in (-0.01, 0.0, +0.01):
        args, extra = an.stigmatic_conic(torch.float64, kappa=an.KAPPA_STAR + dk, pad_rows=2)
        kap = extra["kappa"].clone().requires_grad_(True)
        o = orc.trace_skew_general(*args, kap, extra["poly"], extra["surf_kind"])
        (o[0] ** 2 + o[1] ** 2).sum().backward()
        grads.append(kap.grad[-1].item())
    assert grads[0] < 0 < grads[2] and abs(grads[1]) < 1e-6 * abs(grads[2])


@pytest.mark.parametrize("kappa", [-1.0, 0.0, -0.5])
def test_sag_and_slope_of_conics_against_their_closed_forms(kappa):
    """Paraboloid (kappa = -1): sag = c rho / 2 exactly, d sag / d rho = c / 2; sphere and an ellipsoid likewise against
    the textbook conic."""
    import analytic_asphere as an
    c = 0.08
    rho = torch.linspace(0, 16, 129, dtype=torch.float64)
    sag, dsag, bad = orc._sag_terms(torch.tensor(c, dtype=torch.float64), torch.tensor(kappa, dtype=torch.float64),
                                    torch.zeros(4, dtype=torch.float64), rho)
    assert not bad.any()
    assert np.abs(sag.numpy() - an.conic_sag(c, kappa, np.sqrt(rho.numpy()))).max() < 1e-14
    want_dsag = c / (2 * np.sqrt(1 - (1 + kappa) * c * c * rho.numpy()))
    assert np.abs(dsag.numpy() - want_dsag).max() < 1e-14
    if kappa == -1.0:
        assert np.abs(dsag.numpy() - c / 2).max() < 1e-16
    # the same through the trace: z_RELU returns the sag at the Newton hit, theta_norm the normal's angle
    args, extra, h = an.sag_probe(c, kappa, torch.float64)
    o = orc.trace_skew_general(*args, extra["kappa"], extra["poly"], [1], aggregate=True)
    st = o[7]
    assert np.abs(st["z_RELU"][0].reshape(-1).numpy() - (an.conic_sag(c, kappa, h) + 1.0)).max() < 1e-12
    assert np.abs(st["theta_norm"][0].reshape(-1).numpy()[1:] - an.conic_normal_angle(c, kappa, h)[1:]).max() < 1e-7
