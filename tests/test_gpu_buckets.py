"""
Every compile-time row bucket of the kernels (NS = 4, 8, 12, 16, 20, 24, 32) against the oracle:
strict forward bit-exact, gradients within 1e-5 of the oracle's fp32 (IEEE sqrt) autograd or 2x its
distance from fp64.  Lenses: truncations / extensions of the synthetic 20-row prescription (extra rows
are flat air/air dummies with a small gap, which every ray passes undeviated).
Also: C-ABI error paths and concurrent use from two host threads.
"""
import threading

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


def lens_args(ta, n_rows, n_rays=(16, 16)):
    """Kernel arguments for the first `n_rows` rows of zoom20 (n_rows <= 20) or zoom20 + dummies."""
    from torchoptics_amd import prescriptions as P
    lens, specs, leaves = P.zoom20("cpu", requires_grad=False)
    tr = ta.RayTracer(mode="circular", n_rays=n_rays, rel_fields=(0., 0.6, 1.), wavelengths=("C", "d", "F"),
                      default_device="cpu")
    a = tr.assemble(specs, lens)
    S = a["c"].shape[-1]
    if n_rows <= S:
        for k in ("c", "t", "mu", "mask"):
            a[k] = a[k][..., :n_rows].contiguous()
        if n_rows < S:                      # image plane right behind the last kept row
            a["t"] = a["t"].clone()
            a["t"][..., -1] = 0.5
    else:
        extra = n_rows - S
        t_last = a["t"][..., -1:].clone()
        a["c"] = torch.cat((a["c"], torch.zeros(1, 1, 1, 1, extra)), -1)
        a["mu"] = torch.cat((a["mu"], torch.ones(1, 1, 1, a["mu"].shape[3], extra)), -1)
        a["mask"] = torch.cat((a["mask"], torch.ones(1, 1, 1, 1, extra, dtype=torch.bool)), -1)
        gaps = torch.full((1, 1, 1, 1, extra), 0.05)
        a["t"] = torch.cat((a["t"][..., :-1], gaps, t_last - 0.05 * extra), -1)
    return a


@pytest.mark.parametrize("n_rows", [1, 2, 3, 4, 5, 8, 9, 12, 13, 16, 17, 20, 21, 24, 25, 28, 32])
def test_every_row_bucket_matches_oracle(ta, n_rows):
    from oracle import trace_oracle as orc
    a = lens_args(ta, n_rows)
    order = ("x", "y", "z", "cx", "cy", "c", "t", "mu")
    want = orc.trace_skew(*[a[k] for k in order], a["mask"], ieee_sqrt=True)
    dev = {k: v.to(DEV) for k, v in a.items()}
    lv = {k: dev[k].clone().requires_grad_(True) for k in ("z", "cy", "c", "t", "mu")}
    args = [lv.get(k, dev[k]) for k in order]
    got = ta.trace_skew(*args, dev["mask"], mode="strict")
    for name, g_, w_ in zip(("x", "y", "cx", "cy", "ok", "back"), got, want):
        assert torch.equal(g_.cpu(), w_), f"S={n_rows}: {name} not bit-exact"
    assert got[4].float().mean().item() > 0.5
    ta.compute_rms2d(got[0], got[1], got[4]).backward()
    ref = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        cl = {k: a[k].to(dt).clone().requires_grad_(True) for k in lv}
        o = orc.trace_skew(*[cl.get(k, a[k].to(dt)) for k in order], a["mask"], ieee_sqrt=(dt == torch.float32))
        orc.compute_rms2d(o[0], o[1], o[4]).backward()
        ref[tag] = {k: v.grad for k, v in cl.items()}
    for k in ("c", "t", "mu"):
        g = lv[k].grad.cpu().numpy()
        e32, e64 = rel_l2(g, ref["f32"][k].numpy()), rel_l2(g, ref["f64"][k].numpy())
        noise = rel_l2(ref["f32"][k].numpy(), ref["f64"][k].numpy())
        assert e32 <= 1e-5 or e64 <= 2 * noise + 1e-6, f"S={n_rows} d/d{k}: vs fp32 {e32:.2e}, vs fp64 {e64:.2e}, noise {noise:.2e}"


def test_more_than_32_rows_is_refused(ta):
    a = lens_args(ta, 33)
    order = ("x", "y", "z", "cx", "cy", "c", "t", "mu")
    with pytest.raises(RuntimeError, match="at most 32"):
        ta.trace_skew(*[a[k].to(DEV) for k in order], a["mask"].to(DEV))


def test_cabi_error_codes(ta):
    """Bad arguments come back as negative codes with a message; nothing is launched."""
    import ctypes as C
    from torchoptics_amd import _lib
    lib = _lib.lib()
    p = _lib.tl_problem()
    assert lib.tl_trace_fwd(None, *([None] * 9), None, 0, None) == -1
    assert b"NULL" in lib.tl_last_error()
    p.F, p.P, p.W, p.S, p.device = 1, 64, 1, 40, 0
    assert lib.tl_trace_fwd(C.byref(p), *([None] * 9), None, 0, None) == -1
    assert b"TL_MAX_SURFACES" in lib.tl_last_error()
    p.S = 3
    assert lib.tl_trace_fwd(C.byref(p), *([None] * 9), None, 0, None) == -1       # required pointers are NULL
    buf = torch.zeros(64, device=DEV)
    m8 = torch.ones(8, dtype=torch.uint8, device=DEV)
    for f in ("x_in", "y_in", "z", "cx", "cy", "c", "t", "mu"):
        setattr(p, f, buf.data_ptr())
    p.mask = m8.data_ptr()
    p.xs_p = p.ys_p = 1
    p.mode = 7
    assert lib.tl_trace_fwd(C.byref(p), *([None] * 9), None, 0, None) == -1 and b"mode" in lib.tl_last_error()
    p.mode = 0
    mom = torch.zeros(1, _lib.TL_NMOM, dtype=torch.float64, device=DEV)
    assert lib.tl_trace_fwd(C.byref(p), *([None] * 8), _lib.ptr(mom), None, 0, None) == -3      # workspace too small
    p.surf_kind = m8.data_ptr()                                                     # kappa / poly missing
    assert lib.tl_trace_fwd(C.byref(p), *([None] * 9), None, 0, None) == -1
    p.surf_kind = None
    p.B = -1                                                                        # lens batch: B >= 0, B*F*W <= 65535
    assert lib.tl_trace_fwd(C.byref(p), *([None] * 9), None, 0, None) == -1 and b"B must" in lib.tl_last_error()
    p.B, p.F = 700, 100
    assert lib.tl_trace_fwd(C.byref(p), *([None] * 9), None, 0, None) == -1 and b"65535" in lib.tl_last_error()
    p.B, p.F = 0, 1                                                                 # B = 0 is read as one lens
    assert lib.tl_workspace_bytes(C.byref(p)) > 0
    torch.cuda.synchronize()


def test_empty_pupil(ta):
    """P = 0 (an empty shard): empty per-ray outputs, zero moments, zero gradients, nothing launched."""
    a = {k: v.to(DEV) for k, v in lens_args(ta, 7).items()}
    c = a["c"].clone().requires_grad_(True)
    x0, y0 = a["x"][:, :, :0], a["y"][:, :, :0]
    out = ta.trace_skew(x0, y0, a["z"], a["cx"], a["cy"], c, a["t"], a["mu"], a["mask"])
    assert out[0].shape[2] == 0 and out[4].numel() == 0
    (out[1].sum() + out[0].sum()).backward()
    assert c.grad is not None and not c.grad.any()


def test_two_host_threads_share_the_library(ta):
    """The C ABI keeps no mutable global state: two Python threads trace different lenses at once
    (each on its own stream) and both get the single-thread answer."""
    from oracle import trace_oracle as orc
    order = ("x", "y", "z", "cx", "cy", "c", "t", "mu")
    cases = [lens_args(ta, 8, (32, 32)), lens_args(ta, 20, (32, 32))]
    want = [orc.trace_skew(*[a[k] for k in order], a["mask"], ieee_sqrt=True) for a in cases]
    results, errors = [None, None], []

    def work(i):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                dev = {k: v.to(DEV) for k, v in cases[i].items()}
                for _ in range(20):
                    out = ta.trace_skew(*[dev[k] for k in order], dev["mask"], mode="strict")
                stream.synchronize()
                results[i] = [o.cpu() for o in out]
        except Exception as e:      # noqa: BLE001
            errors.append(e)
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    for got, w in zip(results, want):
        assert all(torch.equal(g_, w_) for g_, w_ in zip(got, w))


def test_plain_cpp_host_on_the_c_abi_matches_python_path(ta):
    """examples/cabi_demo.bin (C++ only: hipMalloc + the tl_* entry points) traces the singlet and
    back-propagates the RMS spot; the Python path on the same fan must give the same numbers."""
    import json
    import os
    import subprocess
    import yaml_free_lenses as L
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "cabi_demo.bin")
    if not os.path.exists(exe):
        from torchoptics_amd.build import build_cabi_demo
        build_cabi_demo(verbose=False)
    out = subprocess.run([exe, "64", "64"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    demo = json.loads(out.stdout.strip().splitlines()[-1])
    lens, specs, leaves = L.build("singlet", DEV)
    tr = ta.RayTracer(mode="circular", n_rays=(64, 64), rel_fields=(0.,), wavelengths=("d",), default_device=DEV)
    a = tr.assemble(specs, lens)
    lv = {k: a[k].detach().clone().requires_grad_(True) for k in ("c", "t", "mu")}
    x, y, cx, cy, ok, back = ta.trace_skew(a["x"], a["y"], a["z"], a["cx"], a["cy"], lv["c"], lv["t"], lv["mu"], a["mask"])
    rms = ta.compute_rms2d(x, y, ok)
    rms.backward()
    assert demo["rays"] == 4096 and demo["ok"] == float(ok.sum().item())
    # the C++ host builds the pupil grid with its own cosf/sinf: equal to ~1e-6 relative, not bit-equal
    assert abs(demo["rms"] - rms.item()) <= 2e-5 * rms.item()
    for k in ("c", "t", "mu"):
        got = np.array(demo["g_" + k])
        assert rel_l2(got, lv[k].grad.cpu().numpy().ravel()) <= 1e-4, k


def test_strided_and_aimed_style_inputs(ta):
    """x_in, y_in arrive as arbitrary strided views (every other pupil point of a larger grid, a full [1,F,P,W] fan stored
    W-major): the kernels read them through element strides; outputs and gradients equal those of contiguous copies."""
    a = {k: v.to(DEV) for k, v in lens_args(ta, 11, n_rays=(16, 32)).items()}
    F, W = a["cy"].shape[1], a["mu"].shape[3]
    big_x = a["x"].expand(1, F, a["x"].shape[2], W).permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1)   # W-major storage
    big_y = a["y"].expand(1, F, a["y"].shape[2], W).permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1)
    xs, ys = big_x[:, :, ::2], big_y[:, :, ::2]                        # every other pupil point: stride 2 over p
    assert not xs.is_contiguous()
    outs, grads = [], []
    for x_in, y_in in ((xs, ys), (xs.contiguous(), ys.contiguous())):
        c = a["c"].clone().requires_grad_(True)
        xg = x_in.clone(memory_format=torch.preserve_format).requires_grad_(True) if x_in.is_contiguous() else None
        o = ta.trace_skew(x_in if xg is None else xg, y_in, a["z"], a["cx"], a["cy"], c, a["t"], a["mu"], a["mask"])
        ta.compute_rms2d(o[0], o[1], o[4]).backward()
        outs.append(o)
        grads.append(c.grad.clone())
    for i in range(6):
        assert torch.equal(outs[0][i], outs[1][i]), i
    assert rel_l2(grads[0].cpu().numpy(), grads[1].cpu().numpy()) < 5e-6      # (per-ray input gradients asked for in run 2: checkpoint kernel there)


def test_many_fields_and_wavelengths_small_pupil(ta):
    """F * W = 600 rows of the launch grid with 64 pupil points each (one wave of four per block holds rays)."""
    from oracle import trace_oracle as orc
    from torchoptics_amd import prescriptions as P
    lens, specs, _ = P.zoom20("cpu", requires_grad=False)
    fields = tuple(np.linspace(0, 1, 200))
    tr = ta.RayTracer(mode="circular", n_rays=(8, 8), rel_fields=fields, wavelengths=("C", "d", "F"), default_device="cpu")
    a = tr.assemble(specs, lens)
    order = ("x", "y", "z", "cx", "cy", "c", "t", "mu")
    want = orc.trace_skew(*[a[k] for k in order], a["mask"], ieee_sqrt=True)
    dev = {k: v.to(DEV) for k, v in a.items()}
    c = dev["c"].clone().requires_grad_(True)
    got = ta.trace_skew(dev["x"], dev["y"], dev["z"], dev["cx"], dev["cy"], c, dev["t"], dev["mu"], dev["mask"])
    assert got[0].shape == (1, 200, 64, 3)
    for i in range(6):
        assert torch.equal(got[i].cpu(), want[i]), i
    ta.compute_rms2d(got[0], got[1], got[4]).backward()
    cc = a["c"].double().clone().requires_grad_(True)
    o = orc.trace_skew(*[(cc if k == "c" else a[k].double()) for k in order], a["mask"])
    orc.compute_rms2d(o[0], o[1], o[4]).backward()
    assert rel_l2(c.grad.cpu().numpy(), cc.grad.numpy()) < 2e-4


def test_non_finite_prescription_behaves_like_the_reference(ta):
    """A NaN curvature / an infinite gap: no hang, no fault; the same rays are flagged as in the oracle and the outputs
    carry NaN in the same places (the reference's comparisons are false for NaN, so such rays are NOT retired)."""
    from oracle import trace_oracle as orc
    a = lens_args(ta, 7)
    order = ("x", "y", "z", "cx", "cy", "c", "t", "mu")
    for key, idx, val in (("c", 2, float("nan")), ("t", 3, float("inf")), ("mu", 1, float("nan"))):
        b = {k: v.clone() for k, v in a.items()}
        b[key][..., idx] = val
        want = orc.trace_skew(*[b[k] for k in order], b["mask"], ieee_sqrt=True)
        got = ta.trace_skew(*[b[k].to(DEV) for k in order], b["mask"].to(DEV))
        assert torch.equal(got[4].cpu(), want[4]) and torch.equal(got[5].cpu(), want[5]), key
        for i in range(4):
            assert torch.equal(torch.isnan(got[i].cpu()), torch.isnan(want[i])), (key, i)
            assert torch.allclose(got[i].cpu(), want[i], rtol=0, atol=0, equal_nan=True), (key, i)


def test_cfg4_total_on_one_gpu(ta):
    """2^27 rays (the whole cfg4 workload) through one launch each way: 64-bit indexing, R = 32 / 64 rays per lane."""
    from torchoptics_amd import prescriptions as P, ray_tracing as rt
    lens, specs, leaves = P.double_gauss(DEV)
    tr = ta.RayTracer(mode="circular", n_rays=(8192, 16384), rel_fields=(0.707,), wavelengths=("d",), default_device=DEV)
    x, y, cx, cy, ok, back = tr.trace_rays(specs, lens)
    assert x.numel() == 1 << 27 and bool(ok.all())
    rms = rt.compute_rms2d(x, y, ok)
    rms.backward()
    tr2 = ta.RayTracer(mode="circular", n_rays=(1024, 1024), rel_fields=(0.707,), wavelengths=("d",), default_device=DEV)
    g_big = leaves["c"].grad.clone()
    leaves["c"].grad = None
    x2, y2, _, _, ok2, _ = tr2.trace_rays(specs, P.double_gauss(DEV)[0].__class__(lens.structure, leaves["c"], leaves["t"], leaves["nd"], leaves["v"]))
    rms2 = rt.compute_rms2d(x2, y2, ok2)
    rms2.backward()
    assert abs(rms.item() - rms2.item()) <= 2e-3 * rms2.item()          # the same spot on a 128 x denser grid
    assert rel_l2(g_big.cpu().numpy(), leaves["c"].grad.cpu().numpy()) < 5e-3
    del x, y, cx, cy, ok, back
    torch.cuda.empty_cache()


def test_forward_without_moments_clears_the_walk_back_flag_word(ta):
    """a C-ABI caller that records tl_trace_fwd WITHOUT moments into a graph has no reduction kernel to clear
    the walk-back's flag word between replays (a step replayed from a graph re-uses its token).  tl_trace_fwd given the full
    tl_workspace_bytes(p) workspace clears the word itself in that case (tl_trace.h, tl_trace_bwd_from_outputs)."""
    import ctypes as C
    from torchoptics_amd import _lib, ops
    from conftest import load_golden
    g = load_golden("G4_tessar_32x32")
    ins = [torch.from_numpy(g[n]).to(DEV) for n in ("in_x", "in_y", "in_z", "in_cx", "in_cy", "in_c", "in_t", "in_mu")]
    mask = torch.from_numpy(g["in_mask"]).to(DEV)
    F, P, W, S = ins[4].shape[1], ins[0].shape[2], ins[7].shape[3], ins[5].shape[-1]
    x_e, y_e = ins[0].expand(1, F, P, W), ins[1].expand(1, F, P, W)
    prob = ops._problem(x_e, y_e, ins[2].reshape(1), ins[3].reshape(1, -1), ins[4].reshape(1, -1).contiguous(),
                        ins[5].reshape(1, S).contiguous(), ins[6].reshape(1, S).contiguous(), ins[7].reshape(1, W, S).contiguous(),
                        mask.reshape(1, S).view(torch.uint8).contiguous(), True, "strict")
    lib = _lib.lib()
    n = lib.tl_workspace_bytes(C.byref(prob))
    ws = torch.full((n,), 0xFF, dtype=torch.uint8, device=DEV)
    outs = [torch.empty((1, F, W, P), dtype=torch.float32, device=DEV) for _ in range(4)]
    flags = [torch.empty((1, F, W, P), dtype=torch.uint8, device=DEV) for _ in range(2)]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.tl_trace_fwd(C.byref(prob), *[_lib.ptr(o) for o in outs], *[_lib.ptr(f) for f in flags], None, None, None,
                          _lib.ptr(ws), ws.numel(), st)
    assert rc == 0, lib.tl_last_error()
    torch.cuda.synchronize()
    word = ws[n - 64:n - 60].cpu().numpy().view(np.uint32)[0]
    assert word == 0 and ws[n - 68].item() == 0xFF and ws[n - 60].item() == 0xFF      # that word, and only it
