"""Analytic cases for the aspheric extension, independent of the oracle AND of the kernels (geometry and Fermat only):
shared by tests/test_oracle_asphere.py (CPU, fp64 / fp32) and tests/test_gpu_asphere.py.

Stigmatic conic (a Cartesian oval for an object at infinity): a single refracting surface between n1 and n2 with conic
constant kappa = -(n1/n2)^2 brings an on-axis collimated beam to a perfect focus at n2 R / (n2 - n1) behind the vertex
-- every ray lands at x = y = 0 there, and by Fermat every ray has the same optical path length from a plane wavefront
to the focus.  sum(x^2 + y^2) at that plane therefore has its minimum, zero, at that kappa: its derivative w.r.t. kappa
changes sign there."""
import numpy as np
import torch

R, N1, N2 = 10.0, 1.0, 1.5
KAPPA_STAR = -(N1 / N2) ** 2
T_FOCUS = N2 * R / (N2 - N1)                       # 30 mm
H_MAX = 5.0                                        # f/2 in air-equivalent terms: (T_FOCUS / N2) / (2 * H_MAX) = 2


PAD_GAP = 0.5                                      # thickness of each flat no-op row put in front of the conic


def stigmatic_conic(dtype=torch.float32, device="cpu", kappa=KAPPA_STAR, n_side=64, z0=-2.0, pad_rows=0):
    """trace_skew arguments of the lens: `pad_rows` flat rows without index change (PAD_GAP apart; they only make the
    lens long enough for the kernels that are built per row count), then the conic; a 64 x 64 square grid of collimated
    on-axis rays clipped to the circle of radius H_MAX, launched from the plane z = z0 in front of the first vertex.
    Returns (args, kwargs); kwargs['surf_kind'] marks the conic row."""
    u = torch.linspace(-H_MAX, H_MAX, n_side, dtype=torch.float64)
    gx, gy = torch.meshgrid(u, u, indexing="ij")
    keep = (gx ** 2 + gy ** 2) <= H_MAX ** 2
    t = lambda v, shape: torch.as_tensor(v, dtype=torch.float64).reshape(shape).to(dtype).to(device)      # noqa: E731
    x, y = t(gx[keep], (1, 1, -1, 1)), t(gy[keep], (1, 1, -1, 1))
    S = pad_rows + 1
    args = [x, y, t(z0, (1, 1, 1, 1)), t(0.0, (1, 1, 1, 1)), t(0.0, (1, 1, 1, 1)),
            t([0.0] * pad_rows + [1.0 / R], (1, 1, 1, 1, S)), t([PAD_GAP] * pad_rows + [T_FOCUS], (1, 1, 1, 1, S)),
            t([1.0] * pad_rows + [N1 / N2], (1, 1, 1, 1, S)), torch.ones(1, 1, 1, 1, S, dtype=torch.bool, device=device)]
    extra = dict(kappa=t([0.0] * pad_rows + [kappa], (S,)), poly=torch.zeros(S, 4, dtype=dtype, device=device),
                 n_index=t([N1] * S + [N2], (1, 1, 1, 1, S + 1)), surf_kind=[0] * pad_rows + [1])
    return args, extra


def expected_opd(z0=-2.0, pad_rows=0):
    """Axial ray: n1 times the way to the conic's vertex + n2 * T_FOCUS behind it."""
    return N1 * (abs(z0) + pad_rows * PAD_GAP) + N2 * T_FOCUS


def sag_probe(c, kappa, dtype=torch.float32, device="cpu", n=257, h_max=4.0, t_shift=-1.0):
    """A one-row 'lens' WITHOUT refraction (mu = 1) hit by axial rays at heights 0..h_max: with thickness t_shift < 0
    the z left behind the row is sag(h^2) - t_shift > 0, so the penalty stack z_RELU of trace_skew(aggregate=True)
    returns the sag itself, and theta_norm the angle between the axis and the surface normal,
    acos(1 / sqrt(1 + (2 h dsag/drho)^2)) / (pi/2)."""
    h = torch.linspace(0.0, h_max, n, dtype=torch.float64)
    t = lambda v, shape: torch.as_tensor(v, dtype=torch.float64).reshape(shape).to(dtype).to(device)      # noqa: E731
    args = [t(h, (1, 1, -1, 1)), t(torch.zeros_like(h), (1, 1, -1, 1)), t(0.0, (1, 1, 1, 1)), t(0.0, (1, 1, 1, 1)),
            t(0.0, (1, 1, 1, 1)), t(c, (1, 1, 1, 1, 1)), t(t_shift, (1, 1, 1, 1, 1)), t(1.0, (1, 1, 1, 1, 1)),
            torch.ones(1, 1, 1, 1, 1, dtype=torch.bool, device=device)]
    extra = dict(kappa=t(kappa, (1,)), poly=torch.zeros(1, 4, dtype=dtype, device=device))
    return args, extra, h.numpy()


def conic_sag(c, kappa, h):
    rho = np.asarray(h, dtype=np.float64) ** 2
    if kappa == -1.0:
        return c * rho / 2                                   # paraboloid: exactly c rho / 2
    return c * rho / (1 + np.sqrt(1 - (1 + kappa) * c * c * rho))


def conic_normal_angle(c, kappa, h):
    """Angle between the axis and the normal of the conic at height h, over pi/2."""
    h = np.asarray(h, dtype=np.float64)
    dsag = c / (2 * np.sqrt(1 - (1 + kappa) * c * c * h * h))  # d sag / d rho
    return np.arccos(1 / np.sqrt(1 + (2 * h * dsag) ** 2)) / (np.pi / 2)
