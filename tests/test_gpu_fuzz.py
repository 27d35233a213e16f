"""
Seeded sweep over the launch shapes of the backward: the default algorithm (walk-back kernels + whatever the checkpoint kernel
queued behind them takes: dead rays of the penalty term, flagged grazing rays, whole fallbacks) against the checkpoint
algorithm run on every ray, on perturbed copies of the reference's own prescriptions.

What varies: the lens (doublet 5 / Cooke 7 / Tessar 8 rows, curvatures and gaps perturbed by up to 3 %), the number of lenses
in the launch (1..4), fields (1..4), wavelengths (1 or 3), the pupil (ragged sizes from 70 to 40 000 points, filled 0.6..1.6 x
the design aperture: overfilled fans have dead and grazing rays), the loss (spot alone, or with the penalty term), aspheric
terms on two rows or none, arithmetic mode.  The two algorithms share no backward code on the rays the walk-back takes, so
agreement to ~1e-5 over 40 such launches is a check of the indexing of every path (chunk plans, scan map, flag bytes, hit
slots, lens batches) rather than of the optics -- those are pinned in the other files.
"""
import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ta():
    import torchoptics_amd
    from torchoptics_amd import _lib
    _lib.lib()
    return torchoptics_amd


def _case(seed):
    import yaml_free_lenses as L
    rng = np.random.default_rng(seed)
    name = ("doublet", "cooke", "tessar")[seed % 3]
    d = L.PRESCRIPTIONS[name]
    S = len(d["c"])
    B = int(rng.integers(1, 5))
    F = int(rng.integers(1, 5))
    W = int(rng.choice([1, 3]))
    P = int(rng.choice([70, 255, 256, 257, 300, 1000, 4097, 40000]))
    fill = float(rng.uniform(0.6, 1.6))
    pen = bool(rng.integers(0, 2))
    asph = bool(rng.integers(0, 2))
    mode = "strict" if rng.integers(0, 3) else "fast"
    c = np.array(d["c"], dtype=np.float32)[None, :] * (1 + 0.03 * rng.standard_normal((B, S))).astype(np.float32)
    t = np.array(d["t"], dtype=np.float32)[None, :] * (1 + 0.03 * rng.random((B, S))).astype(np.float32)
    c[:, d["stop_idx"][0]] = 0.0
    # refractive indices per wavelength, mildly dispersive; mu = n_before / n_after per row
    seq = d["sequence"][0]
    nd = iter(d["nd"])
    n_rows = np.array([next(nd) if ch == "G" else 1.0 for ch in seq], dtype=np.float32)
    lam = np.linspace(-1.0, 1.0, W, dtype=np.float32) if W > 1 else np.zeros(1, dtype=np.float32)
    n = 1.0 + (n_rows[None, :] - 1.0) * (1.0 + 0.01 * lam[:, None])                  # [W,S]
    n = np.concatenate([np.ones((W, 1), np.float32), n], axis=1)
    mu = (n[:, :-1] / n[:, 1:]).astype(np.float32)
    r = np.sqrt(rng.random(P)).astype(np.float32) * (0.5 * L.EPD * fill)
    th = (rng.random(P) * 2 * np.pi).astype(np.float32)
    # (with the penalty term the fields start off axis: on the axis whole rings of rays sit AT the reference's clamp of acos
    #  at 1 - 1e-7, where d theta / d cos^2 jumps from ~1000 to 0 and two fp32 evaluations legitimately disagree by per cent)
    cy = np.sin(np.deg2rad(np.linspace(3.0 if pen else 0.0, L.HFOV_DEG * float(rng.uniform(0.5, 1.3)), F))).astype(np.float32)
    z = float(rng.uniform(2.0, 6.0))
    kap = pol = None
    if asph:
        kap = np.zeros(S, np.float32)
        pol = np.zeros((S, 4), np.float32)
        rows = [0, S - 1]
        kap[rows] = rng.uniform(-0.8, 0.4, 2)
        pol[rows, 0] = rng.uniform(-3e-5, 3e-5, 2)
        pol[rows, 1] = rng.uniform(-3e-7, 3e-7, 2)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)      # noqa: E731
    return dict(name=name, B=B, F=F, W=W, P=P, S=S, fill=fill, pen=pen, asph=asph, mode=mode,
                x=T((r * np.cos(th)).reshape(1, 1, P, 1)), y=T((r * np.sin(th)).reshape(1, 1, P, 1)),
                z=torch.full((B, 1, 1, 1), z, device=DEV), cx=torch.zeros(1, 1, 1, 1, device=DEV), cy=T(cy.reshape(1, F, 1, 1)),
                c=T(c.reshape(B, 1, 1, 1, S)), t=T(t.reshape(B, 1, 1, 1, S)), mu=T(mu.reshape(1, 1, 1, W, S)),
                mask=torch.ones(1, 1, 1, 1, S, dtype=torch.bool, device=DEV),
                kappa=None if kap is None else T(kap), poly=None if pol is None else T(pol))


def _run(ta, k, algo):
    from torchoptics_amd import ops, ray_tracing as rt
    ops.set_backward_algorithm(algo)
    try:
        lv = [k[n].clone().requires_grad_(True) for n in ("c", "t", "mu")]
        kw = {}
        if k["asph"]:
            lv += [k["kappa"].clone().requires_grad_(True), k["poly"].clone().requires_grad_(True)]
            kw.update(kappa=lv[3], poly=lv[4])
        x, y = k["x"], k["y"]
        if k["pen"]:
            x, y = x.expand(1, k["F"], k["P"], k["W"]), y.expand(1, k["F"], k["P"], k["W"])
        out = ta.trace_skew(x, y, k["z"], k["cx"], k["cy"], lv[0], lv[1], lv[2], k["mask"], "sum" if k["pen"] else False, True,
                            mode=k["mode"], **kw)
        if k["B"] > 1:
            ld = rt.unsupervised_loss_batch(out, k["S"], 0.2) if k["pen"] else dict(loss_unsup=rt.compute_rms2d_batch(out[0], out[1], out[4]))
            loss = (ld["loss_unsup"] * torch.linspace(0.5, 1.5, k["B"], device=DEV)).sum()
        else:
            loss = rt.unsupervised_loss(out, k["S"], 0.2)["loss_unsup"] if k["pen"] else ta.compute_rms2d(out[0], out[1], out[4])
        loss.backward()
        mom = out[1]._tl_spot[0]
        stats = dict(rays=out[4].numel(), ok=int(out[4].sum().item()), ill=int(mom[:, 9].sum().item()), inv=ops.used_walk_back(out[0]))
        return [q.grad.clone() for q in lv], stats, float(loss.item())
    finally:
        ops.set_backward_algorithm("inverse")


@pytest.mark.parametrize("seed", list(range(40)))
def test_default_backward_equals_the_checkpoint_algorithm_on_random_launch_shapes(ta, seed):
    k = _case(seed)
    got, st, l1 = _run(ta, k, "inverse")
    want, _, l2 = _run(ta, k, "checkpoint")
    assert l1 == l2                                                     # same forward, same loss
    tag = (f"seed {seed}: {k['name']} B={k['B']} F={k['F']} W={k['W']} P={k['P']} fill={k['fill']:.2f} pen={k['pen']} asph={k['asph']} "
           f"{k['mode']}; {st['ok']}/{st['rays']} live, {st['ill']} flagged, walk-back {st['inv']}")
    if st["ok"] == 0:
        pytest.skip(tag + ": every ray dead")
    names = ("c", "t", "mu") + (("kappa", "poly") if k["asph"] else ())
    errs = []
    for n, a, b in zip(names, got, want):
        assert torch.isfinite(a).all(), (tag, n)
        e = rel_l2(a.cpu().numpy(), b.cpu().numpy()) if float(b.abs().max()) > 0 else float(a.abs().max())
        errs.append(f"{n} {e:.1e}")
        # the penalty term's own fp32 noise is ~1e-4 (d theta / d cos^2 ~ 2000 near normal incidence: test_gpu_penalty.py);
        # the spot loss alone agrees to the walk-back's rounding
        lim = (5e-4 if k["pen"] else 3e-5) * (10.0 if k["mode"] == "fast" else 1.0)
        assert e <= lim, (tag, n, e)
    print(tag + " | " + ", ".join(errs))
