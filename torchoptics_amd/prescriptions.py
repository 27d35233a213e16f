"""
Synthesized lens prescriptions for the BASELINE configs the reference has no data for
(its data/ directory stops at the 8-row Tessar).

Row convention as in the reference YAML files: one row = one surface followed by its medium
('G' glass / 'A' air); c = curvature [1/mm], t = distance to the next vertex [mm]; nd, v only for
glass rows; `stop_idx` = row of the (flat, air/air) aperture stop.

`double_gauss()`  : 11 rows GAGGAAGGAGA (10 refracting surfaces + stop), a classic 6-element
                    f/3 double Gauss form (textbook 100 mm prescription) scaled to EFL = 17.156 mm,
                    the focal length of the reference's own lenses (cfg3 / cfg4).
`zoom20()`        : 20 rows, a double Gauss followed by a weak 4-element relay group (cfg5).
"""
import numpy as np
import torch

from . import lens_modeling as lm

# radii [mm] (0 = flat), thickness [mm], glass (nd, v) of the 100 mm f/3 form
_DG_R = [54.153, 152.522, 35.951, 0.0, 22.270, 0.0, -25.685, 0.0, -36.980, 196.417, -67.148]
_DG_T = [8.747, 0.5, 14.0, 3.777, 14.253, 12.428, 3.777, 10.834, 0.5, 6.858, 57.315]
_DG_SEQ = "GAGGAAGGAGA"
_SK2, _SK16, _F5 = (1.60738, 56.65), (1.62041, 60.32), (1.60342, 38.03)
_DG_GLASS = [_SK2, _SK16, _F5, _F5, _SK16, _SK16]
_DG_EFL = 100.0
TARGET_EFL = 17.15606


def _flat(scale, radii, thick):
    c = [0.0 if r == 0.0 else 1.0 / (r * scale) for r in radii]
    t = [d * scale for d in thick]
    return c, t


# the 2-asphere variant (BASELINE configs[2]): mild conic + 4th/6th-order terms on the rear surface of
# the first element (row 1) and on the last surface (row 10); values chosen so that every ray of the
# f/3, 14-degree fan still passes and the Newton iteration has real work to do (sag departure ~1 um)
_DG_ASPH = {1: (-0.6, (2.0e-6, -4.0e-8, 0.0, 0.0)), 10: (0.4, (-3.0e-6, 5.0e-8, 0.0, 0.0))}
# the same two rows as STRONG aspheres: sag departure from the base sphere 0.32 mm (row 1) and 0.11 mm (row 10) at the
# edge of the f/3 fan, every ray still passes, and the Newton iteration of the forward needs three evaluations per row
# instead of the mild variant's two (measured with the oracle; the image is of course ruined: rms 0.43 mm -- a stress
# prescription for the cost of the iteration, not a design)
_DG_ASPH_STRONG = {1: (-3.0, (1.5e-3, -2.0e-5, 0.0, 0.0)), 10: (2.0, (-1.2e-3, 1.5e-5, 0.0, 0.0))}


def double_gauss(device="cuda", requires_grad=True, dtype=torch.float32, aspheres=False):
    """Returns (lens, specs, leaves) for the 11-row double Gauss; `aspheres=True` makes rows 1 and 10
    aspheric (leaves then also hold kappa [11] and poly [11,4]), `aspheres="strong"` strongly so.
    epd = EFL/3, half field 14 degrees (the form's native aperture and field)."""
    s = TARGET_EFL / _DG_EFL
    c, t = _flat(s, _DG_R, _DG_T)
    asph = None
    if aspheres:
        kap = [0.0] * len(c)
        pol = [[0.0] * 4 for _ in c]
        for row, (k, a) in (_DG_ASPH_STRONG if aspheres == "strong" else _DG_ASPH).items():
            kap[row], pol[row] = k, list(a)
        asph = (kap, pol)
    return _build(_DG_SEQ, 5, c, t, _DG_GLASS, epd=TARGET_EFL / 3.0, hfov_deg=14.0, device=device,
                  requires_grad=requires_grad, dtype=dtype, asph=asph)


def zoom20(device="cuda", requires_grad=True, dtype=torch.float32):
    """20 rows: the double Gauss above (its last gap shortened to 1 mm) + four thin weak elements
    (8 rows GAGAGAGA) + a flat dummy row whose gap is solved for paraxial focus.  A synthetic
    stress prescription for the 20-surface config (cfg5); not a real zoom design."""
    s = TARGET_EFL / _DG_EFL
    c1, t1 = _flat(s, _DG_R, _DG_T)
    t1[-1] = 1.0
    r2 = [90.0, -140.0, -70.0, 110.0, 85.0, -160.0, -95.0, 220.0, 0.0]
    d2 = [0.6, 0.2, 0.6, 0.2, 0.6, 0.2, 0.6, 0.3, 1.0]
    c2 = [0.0 if r == 0.0 else 1.0 / r for r in r2]
    glass = _DG_GLASS + [_SK16, _F5, _SK16, _F5]
    seq = _DG_SEQ + "GAGAGAGAA"
    # solve the last gap so that the image plane sits at the paraxial focus (fp64, host side)
    from . import paraxial
    probe, _, _ = _build(seq, 5, c1 + c2, t1 + d2, glass, 1.0, 1.0, "cpu", False, torch.float64)
    d2[-1] = float(paraxial.get_first_order(probe)[1])
    return _build(seq, 5, c1 + c2, t1 + d2, glass, epd=TARGET_EFL / 4.0, hfov_deg=10.0, device=device,
                  requires_grad=requires_grad, dtype=dtype)


def _build(seq, stop, c, t, glass, epd, hfov_deg, device, requires_grad, dtype, asph=None):
    st = lm.Structure(stop_idx=np.array([stop]), sequence=np.array([seq]), default_device=device)
    mk = lambda v: torch.tensor(v, dtype=dtype, device=device, requires_grad=requires_grad)  # noqa: E731
    leaves = dict(c=mk(c), t=mk(t), nd=mk([g[0] for g in glass]), v=mk([g[1] for g in glass]))
    if asph is not None:
        leaves.update(kappa=mk(asph[0]), poly=mk(asph[1]))
    lens = lm.Lens(st, leaves["c"], leaves["t"], leaves["nd"], leaves["v"], leaves.get("kappa"), leaves.get("poly"))
    specs = lm.Specs(st, torch.tensor([epd], dtype=dtype, device=device),
                     torch.tensor([np.deg2rad(hfov_deg)], dtype=dtype, device=device))
    return lens, specs, leaves
