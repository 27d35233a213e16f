"""
CPU tests of the host-side logic of torchoptics_amd (no kernels involved): lens containers,
dispersion, paraxial utilities, pupil samplers and the assembly of the kernel arguments, all
against fixtures produced by the reference.  Where a test needs a trace (ray aiming, leaf
gradients) the package's `trace_skew` is monkeypatched with the CPU oracle -- the oracle is the
checker here, the product itself has no CPU path (see test_product_has_no_cpu_path).
"""
import re
import os

import numpy as np
import pytest
import torch

import yaml_free_lenses as L
from conftest import ROOT, assert_matches_fixture, load_golden, rel_l2, same_cpu_math
from oracle import trace_oracle as orc

import torchoptics_amd as ta
from torchoptics_amd import lens_modeling as lm, paraxial, ray_tracing as rt

F3 = (0., 0.707, 1.)


def test_samplers_match_reference():
    g = load_golden("G0_samplers")
    xc, yc = rt.circle(None, 8, 8, "cpu")
    xt, yt = rt.tee(None, "cpu")
    assert_matches_fixture(xc.numpy(), g["circle_x"], 2e-7, what="circle x")
    assert_matches_fixture(yc.numpy(), g["circle_y"], 2e-7, what="circle y")
    assert np.array_equal(xt.numpy(), g["tee_x"]) and np.array_equal(yt.numpy(), g["tee_y"])
    torch.manual_seed(0)
    xr, yr = rt.circle_pseudo_random(torch.zeros(1, 1, 1, 1), 8, 8)
    assert_matches_fixture(xr.numpy(), g["rand_x"], 2e-7, what="stratified x")
    assert_matches_fixture(yr.numpy(), g["rand_y"], 2e-7, what="stratified y")


def test_circle_index_range_is_a_slice_of_circle():
    x, y = rt.circle(None, 16, 8, "cpu")
    for a, b in ((0, 128), (5, 77), (100, 128)):
        xs, ys = rt.circle_index_range(16, 8, a, b, "cpu")
        assert torch.equal(xs, x[:, :, a:b]) and torch.equal(ys, y[:, :, a:b])


@pytest.mark.parametrize("name", list(L.PRESCRIPTIONS))
def test_dispersion_and_paraxial(name):
    g8, g9 = load_golden("G8_dispersion"), load_golden("G9_paraxial")
    lens, specs, _ = L.build(name, "cpu", grad=False)
    assert_matches_fixture(lens.get_refractive_indices([656.3, 587.6, 486.1]).numpy(), g8[name + "_n_CdF"], 3e-7, what="n CdF")
    assert_matches_fixture(lens.get_refractive_indices([459., 520., 640.]).numpy(), g8[name + "_n_rgb"], 3e-7, what="n rgb")
    efl, bfl = paraxial.get_first_order(lens)
    pz = paraxial.compute_pupil_position(lens)
    assert_matches_fixture(np.array([efl.item(), bfl.item(), pz.item()]), g9[name], 0, rtol=2e-6, what="first order")
    last = paraxial.compute_last_curvature(lens.structure, lens.flat_c_but_last, lens.flat_t, lens.flat_nd)
    assert_matches_fixture(last.numpy(), g9[name + "_last_c"], 1e-7, rtol=2e-6, what="last curvature")
    assert lens.efl.item() == efl.item() and lens.entrance_pupil_position.item() == pz.item()


def test_glass_variable_round_trip():
    g8 = load_golden("G8_dispersion")
    cat = torch.from_numpy(g8["catalog"])
    g = lm.g_from_n_v(*torch.unbind(cat, dim=1))
    assert_matches_fixture(g.numpy(), g8["catalog_g"], 1e-5, rtol=2e-6, what="catalog g")
    n, v = lm.n_v_from_g(g)
    assert_matches_fixture(n.numpy(), g8["n_back"], 1e-6, what="n back")
    assert_matches_fixture(v.numpy(), g8["v_back"], 1e-4, what="v back")
    near, _ = lm.map_glass_to_closest(g[:5] + 1e-4, g)
    assert torch.equal(near, g[:5])


def test_structure_and_lens_containers():
    lens, specs, leaves = L.build("tessar", "cpu")
    st = lens.structure
    assert st.mask.shape == (1, 8) and st.mask_G.sum() == 4 and len(st) == 1
    assert st.last_g_idx.tolist() == [6] and not st.mask_except_last[0, 7]
    front = lens.up_to_stop()
    assert front.c.shape == (1, 4) and torch.equal(front.flat_c, lens.flat_c[:4])
    assert torch.equal(lens.flat_t, leaves["t"]) and torch.equal(lens.flat_nd, leaves["nd"])
    assert torch.isnan(lens.v[0, 1]) and lens.nd[0, 1] == 1
    half = lens.scale(0.5)
    assert torch.allclose(half.efl, lens.efl * 0.5, rtol=1e-6)
    d = lens.detach()
    assert not d.c.requires_grad and lens.c.requires_grad
    lens.flat_c = torch.arange(8, dtype=torch.float32)
    assert lens.c[0, 3] == 3
    assert lens[0].c.shape == (1, 8) and specs[0].epd.shape == (1,)
    assert lens.double().c.dtype == torch.float64


@pytest.mark.parametrize("case,name,n_rays,wl,epd,hfov", [
    ("G1_singlet_cfg1", "singlet", (64, 64), ("d",), L.EPD, 25.0),
    ("G2_cooke_16x16", "cooke", (16, 16), ("C", "d", "F"), L.EPD, 25.0),
    ("G4_doublet_32x32", "doublet", (32, 32), ("C", "d", "F"), L.EPD, 25.0),
    ("G4_tessar_32x32", "tessar", (32, 32), ("C", "d", "F"), L.EPD, 25.0),
    ("G5_cooke_failures", "cooke", (32, 32), ("C", "d", "F"), 16.0, 35.0),
])
def test_assemble_reproduces_reference_kernel_inputs(case, name, n_rays, wl, epd, hfov):
    g = load_golden(case)
    lens, specs, _ = L.build(name, "cpu", epd=epd, hfov_deg=hfov)
    fields = (0.,) if name == "singlet" else F3
    tr = ta.RayTracer(mode="circular", n_rays=n_rays, rel_fields=fields, wavelengths=wl, default_device="cpu")
    a = tr.assemble(specs, lens)
    for k in ("x", "y", "z", "cx", "cy", "c", "t", "mu", "mask"):
        got = a[k].detach().numpy()
        assert_matches_fixture(got, g["in_" + k], 2e-6, rtol=2e-6, what=f"{case}: in_{k}")


def test_leaf_gradients_through_host_chain(monkeypatch):
    """d rms / d(c, t, nd, v): host chain (this package) + trace (oracle) == reference autograd."""
    monkeypatch.setattr(rt, "trace_skew", lambda *a, mode=None, **k: orc.trace_skew(*a, **k))
    g = load_golden("G2_cooke_16x16")
    lens, specs, leaves = L.build("cooke", "cpu")
    tr = ta.RayTracer(mode="circular", n_rays=(16, 16), rel_fields=F3, wavelengths=("C", "d", "F"), default_device="cpu")
    x, y, cx, cy, ok, back = tr.trace_rays(specs, lens)
    rms = orc.compute_rms2d(x, y, ok)
    assert_matches_fixture(np.float32(rms.item()), np.float32(g["rms"]), 0, rtol=5e-6, what="rms")
    grads = torch.autograd.grad(rms, [leaves[k] for k in ("c", "t", "nd", "v")])
    for k, got in zip(("c", "t", "nd", "v"), grads):
        assert rel_l2(got.numpy(), g["g_" + k]) < (1e-6 if same_cpu_math() else 5e-5), k


def test_ray_aiming_matches_reference(monkeypatch):
    monkeypatch.setattr(rt, "trace_skew", lambda *a, mode=None, **k: orc.trace_skew(*a, **k))
    g, g9 = load_golden("G6_cooke_aim1"), load_golden("G9_paraxial")
    lens, specs, leaves = L.build("cooke", "cpu")
    pr = paraxial.compute_pupil_radius(specs.up_to_stop(), lens.up_to_stop(), default_device="cpu")
    assert_matches_fixture(pr.detach().numpy(), g9["cooke_pupil_radius"], 0, rtol=2e-6, what="pupil radius")
    tr = ta.RayTracer(mode="circular", n_rays=(16, 16), rel_fields=F3, wavelengths=("C", "d", "F"),
                      n_ray_aiming_iter=1, default_device="cpu")
    a = tr.assemble(specs, lens)
    assert a["x"].shape == g["in_x"].shape == (1, 3, 256, 3)
    assert np.abs(a["x"].numpy() - g["in_x"]).max() < 2e-6 and np.abs(a["y"].numpy() - g["in_y"]).max() < 2e-6
    assert not a["x"].requires_grad
    x, y, cx, cy, ok, back = tr.trace_rays(specs, lens)
    rms = orc.compute_rms2d(x, y, ok)
    assert abs(rms.item() - float(g["rms"])) < 2e-7
    grads = torch.autograd.grad(rms, [leaves[k] for k in ("c", "t", "nd")])
    for k, got in zip(("c", "t", "nd"), grads):
        assert rel_l2(got.numpy(), g["g_" + k]) < 2e-4, k


def test_rms_from_moments_equals_oracle():
    g = load_golden("G5_cooke_failures")
    y, ok = torch.from_numpy(g["y"]), torch.from_numpy(g["ok"])
    m = orc.spot_moments(y, ok)
    m8 = torch.zeros(3, 8, dtype=torch.float64)
    m8[:, :4] = m
    got = rt.rms_from_moments(m8, y.shape[2] * y.shape[3])
    assert abs(got.item() - orc.compute_rms2d(None, y.double(), ok).item()) < 1e-13


def test_product_has_no_cpu_path():
    lens, specs, _ = L.build("cooke", "cpu")
    tr = ta.RayTracer(mode="circular", n_rays=(4, 4), default_device="cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tr.trace_rays(specs, lens)
    pkg = os.path.join(ROOT, "torchoptics_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f"{fn} imports the oracle"


def test_unsupported_options_raise():
    with pytest.raises(ValueError):
        ta.RayTracer(mode="bogus")
    assert ta.RayTracer(mode="circular", double_precision=True).double_precision        # round 3: the fp64 kernels


def test_cabi_library_exports_every_declared_symbol():
    """libtltrace.so loads and exports exactly what include/tl_trace.h declares (no compute)."""
    import ctypes
    from torchoptics_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "tl_trace.h")).read()
    declared = set(re.findall(r"\b(tl_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    dll = _lib.lib()
    for name in declared:
        assert hasattr(dll, name), name
    assert dll.tl_version() == int(re.search(r"#define TL_ABI_VERSION (\d+)", hdr).group(1))
    assert dll.tl_last_error() is not None
    p = _lib.tl_problem()
    p.F, p.P, p.W, p.S = 3, 1 << 20, 3, 7
    assert dll.tl_workspace_bytes(ctypes.byref(p)) > 0
    assert ctypes.sizeof(_lib.tl_problem) == dll.tl_problem_size() == 248


def test_new_entry_points_refuse_bad_arguments_before_touching_a_device():
    """tl_unsup_loss / tl_unsup_loss_bwd / tl_aim_fan check their arguments first: TL_EINVAL and a message, no HIP call
    (so this runs without a GPU); the scan map of the penalty walk-back is part of tl_workspace_bytes."""
    import ctypes as C
    from torchoptics_amd import _lib
    dll = _lib.lib()
    one = C.c_void_p(8)                                   # any non-NULL pointer: never dereferenced on these paths
    EINVAL = -1                                           # TL_EINVAL (include/tl_trace.h)
    assert dll.tl_unsup_loss(0, 0, 3, 100.0, one, None, 7.0, 0.2, one, one, one, one, None) == EINVAL
    assert dll.tl_unsup_loss(0, 2, 3, 100.0, one, None, 0.0, 0.2, one, one, one, one, None) == EINVAL   # no n_sequence at all
    assert dll.tl_unsup_loss(0, 2, 3, 100.0, one, None, 7.0, 0.2, None, one, one, one, None) == EINVAL
    assert b"tl_unsup_loss" in dll.tl_last_error()
    assert dll.tl_unsup_loss_bwd(0, 2, 3, one, None, None, None, 1, None, 7.0, 0.2, one, None) == EINVAL   # no upstream gradient
    assert dll.tl_unsup_loss_bwd(0, 2, 3, one, one, None, None, 2, None, 7.0, 0.2, one, None) == EINVAL    # stride 0 or 1
    assert dll.tl_aim_fan(0, 1, 3, 3, 0, one, one, one, one, one, one, one, one, None) == EINVAL
    assert dll.tl_aim_fan(0, 1, 3, 3, 64, one, None, one, one, one, one, one, one, None) == EINVAL
    assert dll.tl_aim_fan(0, 30000, 3, 3, 64, one, one, one, one, one, one, one, one, None) == EINVAL      # grid rows
    p = _lib.tl_problem()
    p.F, p.W, p.S = 1, 1, 11
    p.P = 1 << 20
    small = dll.tl_workspace_bytes(C.byref(p))
    p.P = 1 << 24
    big = dll.tl_workspace_bytes(C.byref(p))
    assert big - small >= ((1 << 24) - (1 << 20)) // 256 * 4          # one byte per chunk and wave of the block


def test_gradient_free_conversions_are_memoised_safely():
    """lens_modeling caches the padding of constant nd / v and their dispersion (an optimisation loop rebuilds
    the same Lens every step).  The cache must notice new data at a recycled address, in-place updates of the
    source, and writes into a cached result."""
    import gc
    from torchoptics_amd import lens_modeling as lm
    st = lm.Structure(stop_idx=np.array([2]), sequence=np.array(["GAAGA"]), default_device="cpu")
    c, t = torch.zeros(5), torch.ones(5)

    def indices(nd_vals, v_vals):
        lens = lm.Lens(st, c, t, torch.tensor(nd_vals), torch.tensor(v_vals))
        return lens.get_refractive_indices((656.3, 587.6, 486.1)).clone()

    a = indices([1.5, 1.6], [60.0, 40.0])
    gc.collect()                                        # the first tensors are gone: their addresses may be reused
    b = indices([1.7, 1.8], [30.0, 50.0])
    assert not torch.equal(a, b) and abs(b[0, 0, 1].item() - 1.7) < 1e-4 and abs(a[0, 0, 1].item() - 1.5) < 1e-4
    nd, v = torch.tensor([1.5, 1.6]), torch.tensor([60.0, 40.0])
    n1 = lm.Lens(st, c, t, nd, v).get_refractive_indices((587.6,))
    n2 = lm.Lens(st, c, t, nd, v).get_refractive_indices((587.6,))
    assert n2.data_ptr() == n1.data_ptr() or torch.equal(n1, n2)            # served from the cache (or equal)
    nd.mul_(1.01)                                                           # in-place update of the source
    n3 = lm.Lens(st, c, t, nd, v).get_refractive_indices((587.6,))
    assert abs(n3[0, 0, 0].item() - 1.515) < 1e-4
    n3.zero_()                                                              # somebody scribbles on a cached result
    n4 = lm.Lens(st, c, t, nd, v).get_refractive_indices((587.6,))
    assert abs(n4[0, 0, 0].item() - 1.515) < 1e-4
    nd_g = nd.clone().requires_grad_(True)                                  # with autograd: always fresh, with a graph
    n5 = lm.Lens(st, c, t, nd_g, v).get_refractive_indices((587.6,))
    assert n5.requires_grad


def test_memo_key_tells_aliasing_views_and_fields_apart():
    """the cache key ignored strides and the field name.  Two gradient-free views with the same
    first element and shape but different strides (a row and a column of one matrix) are different data, and c
    and t padded with the same fill must never share an entry."""
    from torchoptics_amd import lens_modeling as lm
    # two lenses of different length, so that padding really happens (5 and 3 rows)
    st = lm.Structure(stop_idx=np.array([2, 1]), sequence=np.array(["GAAGA", "AGA"]), default_device="cpu")
    M = torch.arange(64, dtype=torch.float32).reshape(8, 8) + 1.0
    row, col = M[0, :], M[:, 0]                        # same data_ptr, same shape [8], strides 1 vs 8
    assert row.data_ptr() == col.data_ptr() and row.shape == col.shape
    nd, v = torch.tensor([1.5, 1.6, 1.7]), torch.tensor([60.0, 40.0, 50.0])
    a = lm.Lens(st, row, row, nd, v)
    b = lm.Lens(st, col, col, nd, v)
    assert torch.equal(a.c[0], row[:5]) and torch.equal(a.c[1, :3], row[5:8])
    assert torch.equal(b.c[0], col[:5]) and torch.equal(b.c[1, :3], col[5:8])
    assert not torch.equal(a.c, b.c)
    # c and t from DIFFERENT tensors that happen to be padded alike: each gets its own values
    cc, tt = torch.full((8,), 2.0), torch.full((8,), 3.0)
    lens = lm.Lens(st, cc, tt, nd, v)
    assert (lens.c[0] == 2.0).all() and (lens.t[0] == 3.0).all() and lens.c[1, 3:].eq(0).all()
