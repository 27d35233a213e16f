#!/usr/bin/env python3
"""
Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Runs only in the build container, where the reference checkout is mounted read-only at
/root/reference.  Nothing from the reference travels: the outputs are plain arrays
(inputs + expected outputs) in .npz files plus scalars in golden_scalars.json.

    python tests/golden/make_golden.py            # regenerates every fixture

The reference imports one module it never uses (`shapely`, ray_tracing_lite.py:15) and
one plotting helper that is missing from its own tree (`utils.w2rgb`,
optics_simulator_lite.py:10); both are satisfied with empty stand-in modules here, as
recorded in SURVEY.md Appendix C.  Fixture list: SURVEY.md Appendix C, G1-G10.
"""
import json
import os
import sys
import types

import numpy as np
import torch
import yaml

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

sys.dont_write_bytecode = True
_sh, _geo = types.ModuleType("shapely"), types.ModuleType("shapely.geometry")
_geo.Polygon = object
_sh.geometry = _geo
sys.modules["shapely"], sys.modules["shapely.geometry"] = _sh, _geo
_u, _w = types.ModuleType("utils"), types.ModuleType("utils.w2rgb")
_w.wavelength_to_rgb = lambda wl: (0, 0, 0)
_u.w2rgb = _w
sys.modules["utils"], sys.modules["utils.w2rgb"] = _u, _w
sys.path.insert(0, REF)

import torchlens.ray_tracing_lite as rt      # noqa: E402
import torchlens.lens_modeling as lm          # noqa: E402
import torchlens.optics_simulator_lite as osl  # noqa: E402

DATA = os.path.join(REF, "torchlens", "data")
LENSES = {"singlet": "singlet_lens.yml", "doublet": "baseline_doublet.yml",
          "cooke": "baseline_cooke.yml", "tessar": "baseline_tessar.yml"}
EPD = 8.57803
HFOV_DEG = 25.0
SCALARS = {}


def load(name):
    with open(os.path.join(DATA, LENSES[name])) as f:
        return yaml.safe_load(f)


def build(name, dtype=torch.float32, grad=True):
    d = load(name)
    st = lm.Structure(stop_idx=np.array(d["stop_idx"]), sequence=np.array(d["sequence"]),
                      default_device="cpu")
    leaves = {k: torch.tensor(d[k], dtype=torch.float32).to(dtype).requires_grad_(grad)
              for k in ("c", "t", "nd", "v")}
    lens = lm.Lens(st, leaves["c"], leaves["t"], leaves["nd"], leaves["v"])
    return d, st, lens, leaves


def specs_for(st, epd=EPD, hfov_deg=HFOV_DEG, dtype=torch.float32):
    return lm.Specs(st, torch.tensor([epd], dtype=torch.float32).to(dtype),
                    torch.tensor([np.deg2rad(hfov_deg)], dtype=torch.float32).to(dtype))


class Capture:
    """Records the arguments the reference hands to its own trace_skew."""

    def __init__(self):
        self.args = None
        self._orig = rt.trace_skew

    def __enter__(self):
        def spy(*a, **k):
            # keep the LAST top-level call (ray aiming makes nested calls first)
            self.args = [v.detach().clone() if torch.is_tensor(v) else v for v in a]
            return self._orig(*a, **k)
        rt.trace_skew = spy
        return self

    def __exit__(self, *exc):
        rt.trace_skew = self._orig


def np32(t):
    return t.detach().cpu().numpy()


def grads_of(loss, leaves):
    gs = torch.autograd.grad(loss, [leaves[k] for k in ("c", "t", "nd", "v")], allow_unused=True)
    return {"g_" + k: (np32(g) if g is not None else np.zeros_like(np32(leaves[k])))
            for k, g in zip(("c", "t", "nd", "v"), gs)}


def input_grads(args, dtype, suffix, keep_rays):
    """d rms / d(trace_skew inputs), from the reference's own trace_skew + compute_rms2d
    evaluated on the captured inputs (cast to `dtype`)."""
    names = ("x", "y", "z", "cx", "cy", "c", "t", "mu")
    ins = [a.to(dtype).clone().requires_grad_(True) for a in args[:8]]
    x, y, ocx, ocy, ok, back = rt.trace_skew(*ins, args[8], False, args[10])
    rms = rt.compute_rms2d(x, y, ok)
    gs = torch.autograd.grad(rms, ins, allow_unused=True)
    out = {"rms_in" + suffix: np.float64(rms.item())}
    for n, g, a in zip(names, gs, ins):
        if n in ("x", "y") and not keep_rays:
            continue
        out["gin_" + n + suffix] = (g if g is not None else torch.zeros_like(a)).detach().numpy()
    return out


def trace_case(name, n_rays, fields, wl, aim=0, allow_back=True, epd=EPD, hfov_deg=HFOV_DEG,
               keep_rays=True):
    """fp32 run through the reference's own RayTracer; returns dict of arrays."""
    d, st, lens, leaves = build(name)
    specs = specs_for(st, epd, hfov_deg)
    tr = rt.RayTracer(mode="circular", n_rays=n_rays, rel_fields=fields, wavelengths=wl,
                      n_ray_aiming_iter=aim, allow_backward_rays=allow_back, default_device="cpu")
    with Capture() as cap:
        x, y, cx, cy, ok, back = tr.trace_rays(specs, lens)
    rms = rt.compute_rms2d(x, y, ok)
    out = {"rms": np.float64(rms.item())}
    out.update(grads_of(rms, leaves))
    names = ("in_x", "in_y", "in_z", "in_cx", "in_cy", "in_c", "in_t", "in_mu", "in_mask")
    for k, v in zip(names, cap.args[:9]):
        out[k] = np32(v)
    out["allow_backward_rays"] = np.bool_(allow_back)
    out.update(input_grads(cap.args, torch.float32, "", keep_rays))
    out.update(input_grads(cap.args, torch.float64, "64", keep_rays))
    out["ok_frac"] = np32(ok.float().mean(dim=(0, 2, 3)))
    out["back_frac"] = np.float64(back.float().mean().item())
    if keep_rays:
        out.update(x=np32(x), y=np32(y), cx=np32(cx), cy=np32(cy), ok=np32(ok), back=np32(back))
    else:
        out.pop("in_x"), out.pop("in_y")
    return out


def trace_case_f64(name, n_rays, fields, wl, epd=EPD, hfov_deg=HFOV_DEG):
    """fp64 'truth' for the same fp32-valued inputs, calling the reference FUNCTIONS
    directly (RayTracer(double_precision=True) is broken)."""
    torch.set_default_dtype(torch.float64)
    try:
        d, st, lens, leaves = build(name, dtype=torch.float64)
        specs = specs_for(st, epd, hfov_deg, dtype=torch.float64)
        conv = {"C": 656.3, "d": 587.6, "F": 486.1}
        wls = [conv.get(w, w) for w in wl]
        n = lens.get_refractive_indices(wls)
        n = torch.cat((torch.ones_like(n[:, 0:1, :]), n), dim=1)
        n = torch.transpose(n, 1, 2)
        n = torch.reshape(n, (n.shape[0], 1, 1, n.shape[1], -1))
        z = torch.reshape(rt.compute_pupil_position(lens), (-1, 1, 1, 1))
        xr, yr = rt.circle(z, *n_rays, "cpu")
        xp = rt.scale_to_epd(xr.double(), specs.epd)
        yp = rt.scale_to_epd(yr.double(), specs.epd)
        f32 = torch.tensor(fields, dtype=torch.float32).double()
        cy = torch.sin((specs.hfov[:, None] * f32[None, :])[..., None, None])
        cx = torch.zeros(1, 1, 1, 1)
        c = torch.reshape(lens.c, (1, 1, 1, 1, -1))
        t = torch.reshape(lens.t, (1, 1, 1, 1, -1))
        mu = n[..., :-1] / n[..., 1:]
        mask = torch.reshape(st.mask_torch, (1, 1, 1, 1, -1))
        x, y, ocx, ocy, ok, back = rt.trace_skew(xp, yp, z, cx, cy, c, t, mu, mask)
        rms = rt.compute_rms2d(x, y, ok)
        out = {"rms64": np.float64(rms.item())}
        for k, v in grads_of(rms, leaves).items():
            out[k + "64"] = v.astype(np.float64)
        yd, okd = y[0].detach(), ok[0].double()
        out["moments64"] = np.stack([yd.sum(dim=(1, 2)).numpy(), (okd * yd).sum(dim=(1, 2)).numpy(),
                                     (okd * yd * yd).sum(dim=(1, 2)).numpy(), okd.sum(dim=(1, 2)).numpy()], 1)
        return out
    finally:
        torch.set_default_dtype(torch.float32)


def save(name, arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def first_order(name):
    d, st, lens, leaves = build(name, grad=False)
    efl, bfl = rt.get_first_order(lens)
    pz = rt.compute_pupil_position(lens)
    return float(efl[0]), float(bfl[0]), float(pz[0])


def batch_case(names, n_rays, fields, wl):
    """G11: a padded BATCH of lenses (B = len(names)) through the reference's own RayTracer / trace_skew in one call --
    the broadcast over dim 0 that its containers exist for.  Records the captured trace_skew inputs [B,...], the
    outputs [B,F,P,W], the penalty stacks (aggregate=True), and d(sum_b rms_b)/d(inputs) from the reference's autograd
    in fp32 and fp64 (rms_b = compute_rms2d on lens b's slice: the reference's compute_rms2d reads sample 0 only)."""
    ds = [load(n) for n in names]
    st = lm.Structure(stop_idx=np.array([d["stop_idx"][0] for d in ds]), sequence=np.array([d["sequence"][0] for d in ds]),
                      default_device="cpu")
    leaves = {k: torch.tensor(sum((d[k] for d in ds), []), dtype=torch.float32) for k in ("c", "t", "nd", "v")}
    lens = lm.Lens(st, leaves["c"], leaves["t"], leaves["nd"], leaves["v"])
    B = len(names)
    specs = lm.Specs(st, torch.tensor([EPD] * B, dtype=torch.float32), torch.tensor([np.deg2rad(HFOV_DEG)] * B, dtype=torch.float32))
    tr = rt.RayTracer(mode="circular", n_rays=n_rays, rel_fields=fields, wavelengths=wl, default_device="cpu")
    with Capture() as cap:
        x, y, cx, cy, ok, back = tr.trace_rays(specs, lens)
    out = dict(x=np32(x), y=np32(y), cx=np32(cx), cy=np32(cy), ok=np32(ok), back=np32(back))
    names_in = ("in_x", "in_y", "in_z", "in_cx", "in_cy", "in_c", "in_t", "in_mu", "in_mask")
    for k, v in zip(names_in, cap.args[:9]):
        out[k] = np32(v)
    # aggregate=True needs full-shape inputs: with the un-aimed [1,1,P,1] pupil grid and W > 1 the reference's own
    # `theta[~ray_ok] = 1.` (ray_tracing_lite.py:653) raises a shape error, for any B.  Hand it the expanded grid.
    full = [cap.args[0].expand_as(x).clone(), cap.args[1].expand_as(x).clone()] + list(cap.args[2:9])
    res = rt.trace_skew(*full, True, True)
    for i, key in enumerate(("x", "y", "cx", "cy", "ok", "back")):
        assert np.array_equal(np32(res[i]), out[key]), key          # same rays as the un-expanded call
    for key, lst in res[6].items():
        out["stack_" + key] = np32(torch.stack(lst, 0))
    for dtype, suffix in ((torch.float32, ""), (torch.float64, "64")):
        ins = [a.to(dtype).clone().requires_grad_(True) for a in cap.args[:8]]
        xo, yo, _, _, oko, _ = rt.trace_skew(*ins, cap.args[8], False, True)
        rms_b = torch.stack([rt.compute_rms2d(xo[b:b + 1], yo[b:b + 1], oko[b:b + 1]) for b in range(B)])
        gs = torch.autograd.grad(rms_b.sum(), ins, allow_unused=True)
        out["rms_b" + suffix] = rms_b.detach().numpy().astype(np.float64)
        for n, g, a in zip(("x", "y", "z", "cx", "cy", "c", "t", "mu"), gs, ins):
            if n not in ("x", "y"):
                out["gin_" + n + suffix] = (g if g is not None else torch.zeros_like(a)).detach().numpy()
    return out


def main():
    F3 = (0., 0.707, 1.)
    CDF = ("C", "d", "F")
    if "--only-G11" in sys.argv:
        save("G11_batch3_16x16", batch_case(("cooke", "doublet", "tessar"), (16, 16), F3, CDF))
        return

    # G1  singlet cfg1
    g1 = trace_case("singlet", (64, 64), (0.,), ("d",))
    efl, bfl, pz = first_order("singlet")
    g1.update(efl=np.float64(efl), bfl=np.float64(bfl), pupil_z=np.float64(pz))
    save("G1_singlet_cfg1", g1)

    # G2  Cooke 16x16 F3 W3, aim 0, fp32 + fp64
    g2 = trace_case("cooke", (16, 16), F3, CDF)
    g2.update(trace_case_f64("cooke", (16, 16), F3, CDF))
    save("G2_cooke_16x16", g2)

    # G3  Cooke 1024x1024 (cfg2) scalars only
    for tag, wl in (("d", ("d",)), ("CdF", CDF)):
        g3 = trace_case("cooke", (1024, 1024), F3, wl, keep_rays=False)
        g3.update(trace_case_f64("cooke", (1024, 1024), F3, wl))
        save("G3_cooke_cfg2_" + tag, g3)

    # G4  doublet, tessar 32x32
    for nm in ("doublet", "tessar"):
        g4 = trace_case(nm, (32, 32), F3, CDF)
        g4.update(trace_case_f64(nm, (32, 32), F3, CDF))
        efl, bfl, pz = first_order(nm)
        g4.update(efl=np.float64(efl), bfl=np.float64(bfl), pupil_z=np.float64(pz))
        save("G4_" + nm + "_32x32", g4)

    # G5  failure-heavy Cooke
    g5 = trace_case("cooke", (32, 32), F3, CDF, epd=16.0, hfov_deg=35.0)
    g5.update(trace_case_f64("cooke", (32, 32), F3, CDF, epd=16.0, hfov_deg=35.0))
    save("G5_cooke_failures", g5)

    # G6  Cooke with one ray-aiming iteration
    g6 = trace_case("cooke", (16, 16), F3, CDF, aim=1)
    save("G6_cooke_aim1", g6)

    # G11 a padded batch of three lenses in one trace_skew call
    save("G11_batch3_16x16", batch_case(("cooke", "doublet", "tessar"), (16, 16), F3, CDF))

    # G10 allow_backward_rays=False
    for nm in ("cooke", "tessar"):
        save("G10_" + nm + "_noback", trace_case(nm, (32, 32), F3, CDF, allow_back=False))

    # G7  harness: RaytracedOptics.do_ray_tracing (rms + penalty), aim=1, aggregate=True
    d = load("cooke")
    leaves = {k: torch.tensor(d[k], dtype=torch.float32, requires_grad=True) for k in ("c", "t", "nd", "v")}
    sim = osl.RaytracedOptics(
        "", stop_index=np.array(d["stop_idx"]), sequence=np.array(d["sequence"]),
        hfov=torch.tensor([0., 17.5, 25.]), epd=torch.tensor([8.578]),
        curvature=leaves["c"], thickness=leaves["t"], n_refractive=leaves["nd"], abbe_number=leaves["v"],
        n_sampled_fields=3, n_pupil_rings=8, pupil_sampling="circular", wavelengths=[459., 520., 640.],
        penalty_rate=0.2, n_ray_aiming_iter=1, lazy_init=True,
        glass_catalog_path=os.path.join(DATA, "selected_ohara_glass.csv"), device="cpu")
    with Capture() as cap:
        x, y, ok = sim.do_ray_tracing(sim.lensR)
    ld = sim.loss_dict
    g7 = {k: np.float64(v.item()) for k, v in ld.items()}
    for key in ("loss_unsup", "rms", "penalty"):
        gs = torch.autograd.grad(ld[key], [leaves[k] for k in ("c", "t", "nd", "v")],
                                 retain_graph=True, allow_unused=True)
        for k, g in zip(("c", "t", "nd", "v"), gs):
            g7[f"g_{key}_{k}"] = np32(g)
    names = ("in_x", "in_y", "in_z", "in_cx", "in_cy", "in_c", "in_t", "in_mu", "in_mask")
    for k, v in zip(names, cap.args[:9]):
        g7[k] = np32(v)
    g7.update(x=np32(x), y=np32(y), ok=np32(ok), n_sequence=np.int64(len(d["sequence"][0])))
    # stacks straight from the reference tracer on the captured inputs
    res = rt.trace_skew(*cap.args[:9], True, True)
    for key, lst in res[6].items():
        g7["stack_" + key] = np32(torch.stack(lst, 0))
    save("G7_harness_cooke", g7)

    # G8  dispersion + glass transforms
    g8 = {}
    for nm in LENSES:
        d, st, lens, _ = build(nm, grad=False)
        g8[nm + "_n_CdF"] = np32(lens.get_refractive_indices([656.3, 587.6, 486.1]))
        g8[nm + "_n_rgb"] = np32(lens.get_refractive_indices([459., 520., 640.]))
    cat = torch.tensor(np.loadtxt(os.path.join(DATA, "selected_ohara_glass.csv"), delimiter=",", dtype=np.float32))
    g = lm.g_from_n_v(*torch.unbind(cat, dim=1))
    n_back, v_back = lm.n_v_from_g(g)
    g8.update(catalog=np32(cat), catalog_g=np32(g), n_back=np32(n_back), v_back=np32(v_back))
    save("G8_dispersion", g8)

    # G9  paraxial utilities
    g9 = {}
    for nm in LENSES:
        d, st, lens, _ = build(nm, grad=False)
        efl, bfl, pz = first_order(nm)
        last_c = rt.compute_last_curvature(st, lens.flat_c_but_last, lens.flat_t, lens.flat_nd)
        g9[nm] = np.array([efl, bfl, pz], dtype=np.float64)
        g9[nm + "_last_c"] = np32(last_c)
        specs2 = specs_for(st).up_to_stop()
        if d["stop_idx"][0] > 0:
            pr = rt.compute_pupil_radius(specs2, lens.up_to_stop(), default_device="cpu")
            g9[nm + "_pupil_radius"] = np32(pr)
    save("G9_paraxial", g9)

    # pupil samplers (deterministic ones) + the seeded stratified one
    xc, yc = rt.circle(None, 8, 8, "cpu")
    xt, yt = rt.tee(None, "cpu")
    torch.manual_seed(0)
    xr, yr = rt.circle_pseudo_random(torch.zeros(1, 1, 1, 1), 8, 8)
    save("G0_samplers", dict(circle_x=np32(xc), circle_y=np32(yc), tee_x=np32(xt), tee_y=np32(yt),
                             rand_x=np32(xr), rand_y=np32(yr)))

    SCALARS.update(torch_version=torch.__version__, note="generated by tests/golden/make_golden.py "
                   "from the reference imported at /root/reference on CPU")
    with open(os.path.join(OUT, "golden_scalars.json"), "w") as f:
        json.dump(SCALARS, f, indent=1)


if __name__ == "__main__":
    main()
