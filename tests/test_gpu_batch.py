"""B > 1: padded batches of lenses of different length (the reference's containers exist for this; its `trace_skew`
broadcasts over dim 0).  ONE launch each way for the whole batch (tl_problem.B): the strict forward equals the IEEE
oracle on the padded batch and the reference's own batch run (fixture G11), per-lens gradients equal the reference's
autograd, compute_rms2d reads lens 0 only, compute_rms2d_batch gives every lens its own spot size."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _batch(device, dtype=torch.float32):
    import yaml_free_lenses as L
    from torchoptics_amd import lens_modeling as lm
    a, b = L.PRESCRIPTIONS["cooke"], L.PRESCRIPTIONS["doublet"]
    st = lm.Structure(stop_idx=np.array(a["stop_idx"] + b["stop_idx"]), sequence=np.array(a["sequence"] + b["sequence"]),
                      default_device=device)
    leaves = {k: torch.tensor(a[k] + b[k], dtype=dtype, device=device, requires_grad=True) for k in ("c", "t", "nd", "v")}
    lens = lm.Lens(st, leaves["c"], leaves["t"], leaves["nd"], leaves["v"])
    specs = lm.Specs(st, torch.tensor([L.EPD, L.EPD], dtype=dtype, device=device),
                     torch.tensor([np.deg2rad(L.HFOV_DEG)] * 2, dtype=dtype, device=device))
    return lens, specs, leaves


def test_two_lens_padded_batch():
    import yaml_free_lenses as L
    import torchoptics_amd as ta
    from oracle import trace_oracle as orc
    lens, specs, leaves = _batch(DEV)
    assert lens.c.shape == (2, 7) and not lens.structure.mask[1, 5:].any()
    kw = dict(mode="circular", n_rays=(16, 16), rel_fields=(0., 0.707, 1.), wavelengths=("C", "d", "F"))
    tr = ta.RayTracer(default_device=DEV, **kw)
    a = tr.assemble(specs, lens)
    out = ta.trace_skew(a["x"], a["y"], a["z"], a["cx"], a["cy"], a["c"], a["t"], a["mu"], a["mask"])
    assert out[0].shape == (2, 3, 256, 3) and out[4].dtype == torch.bool
    # (1) the strict forward on the padded batch = the IEEE oracle on the same padded batch
    cpu = {k: v.detach().cpu() for k, v in a.items()}
    want = orc.trace_skew(cpu["x"], cpu["y"], cpu["z"], cpu["cx"], cpu["cy"], cpu["c"], cpu["t"], cpu["mu"], cpu["mask"],
                          ieee_sqrt=True)
    for i in range(6):
        assert torch.equal(out[i].cpu(), want[i]), i
    # (2) each lens of the batch = that lens traced alone (tolerances from round 2's first half, when a batch took the
    #     fp32 ABCD chain for z; now tl_pupil_position serves batches too)
    for b, name in enumerate(("cooke", "doublet")):
        l1, s1, _ = L.build(name, DEV)
        o1 = tr.trace_rays(s1, l1)
        for i in range(4):
            assert (out[i][b:b + 1] - o1[i]).abs().max().item() < 5e-6, (name, i)
        assert torch.equal(out[4][b:b + 1], o1[4]) and torch.equal(out[5][b:b + 1], o1[5])
    # (3) compute_rms2d reads sample 0 (ray_tracing_lite.py:695,699) and back-propagates to lens 0's rows only;
    #     a loss on lens 1's rays reaches lens 1's rows
    l0, s0, lv0 = L.build("cooke", DEV)
    o0 = tr.trace_rays(s0, l0)
    rms0 = ta.compute_rms2d(o0[0], o0[1], o0[4])
    rms = ta.compute_rms2d(out[0], out[1], out[4])
    assert abs(rms.item() - rms0.item()) <= 1e-5 * rms0.item()
    (rms + out[1][1].square().mean()).backward()
    rms0.backward()
    g = leaves["c"].grad
    assert torch.allclose(g[:7], lv0["c"].grad, rtol=2e-3, atol=1e-6) and g[7:].abs().max().item() > 0


def _g11(device):
    from conftest import load_golden
    g = load_golden("G11_batch3_16x16")
    names = ("in_x", "in_y", "in_z", "in_cx", "in_cy", "in_c", "in_t", "in_mu")
    return g, [torch.from_numpy(g[n]).to(device) for n in names], torch.from_numpy(g["in_mask"]).to(device)


@pytest.mark.parametrize("mode", ["strict", "fast"])
def test_reference_batch_fixture_forward_and_stacks(mode):
    """G11 = the reference's own trace of three padded lenses in one call: rays, masks, penalty stacks."""
    import torchoptics_amd as ta
    from oracle import trace_oracle as orc
    g, ins, mask = _g11(DEV)
    out = ta.trace_skew(*ins, mask, mode=mode)
    tol = 1e-5 if mode == "strict" else 3e-5
    for i, name in enumerate(("x", "y", "cx", "cy")):
        assert np.abs(out[i].cpu().numpy() - g[name]).max() <= (tol if i < 2 else tol / 10), name
    assert np.array_equal(out[4].cpu().numpy(), g["ok"]) and np.array_equal(out[5].cpu().numpy(), g["back"])
    if mode == "strict":        # bit-exact with the correctly rounded evaluation of the same op sequence
        want = orc.trace_skew(*[a.cpu() for a in ins], mask.cpu(), ieee_sqrt=True)
        assert all(torch.equal(out[i].cpu(), want[i]) for i in range(6))
    agg = ta.trace_skew(*ins, mask, aggregate=True, mode=mode)
    for key in ("z_RELU", "theta_norm", "theta_prime_norm"):
        got = torch.stack(agg[6][key], 0).cpu().numpy()
        assert got.shape == g["stack_" + key].shape
        assert np.abs(got - g["stack_" + key]).max() <= (2e-4 if key != "z_RELU" else 2e-5), key
    # the fused penalty sum per lens = the sum over that lens' stacks
    q = sum(np.nan_to_num(g["stack_" + k].astype(np.float64)).sum(axis=0) for k in ("z_RELU", "theta_norm", "theta_prime_norm"))
    assert np.allclose(agg[6].q_per_lens.cpu().numpy(), q.sum(axis=(1, 2, 3)), rtol=2e-5)


@pytest.mark.parametrize("algo", ["inverse", "checkpoint"])
def test_reference_batch_fixture_gradients(algo):
    """d(sum_b rms_b)/d(z, cy, c, t, mu) per lens against the reference's autograd (fp32 and fp64 runs in G11)."""
    import torchoptics_amd as ta
    from conftest import rel_l2
    from torchoptics_amd import ops
    g, ins, mask = _g11(DEV)
    lv = [a.clone().requires_grad_(True) for a in ins[2:]]
    ops.set_backward_algorithm(algo)
    try:
        out = ta.trace_skew(ins[0], ins[1], *lv, mask)
        rms_b = ta.compute_rms2d_batch(out[0], out[1], out[4])
        assert np.allclose(rms_b.detach().cpu().numpy(), g["rms_b64"], rtol=2e-5)
        assert abs(ta.compute_rms2d(out[0], out[1], out[4]).item() - g["rms_b64"][0]) <= 2e-5 * g["rms_b64"][0]
        rms_b.sum().backward()
    finally:
        ops.set_backward_algorithm("inverse")
    for n, q in zip(("z", "cx", "cy", "c", "t", "mu"), lv):
        got, w32, w64 = q.grad.cpu().numpy(), g["gin_" + n], g["gin_" + n + "64"]
        assert got.shape == w64.shape
        if n == "cx":
            continue                                        # zero by symmetry: noise over noise
        noise = rel_l2(w32, w64)
        lim = 3e-5 if n in ("c", "t", "mu") else 1e-3       # z, cy: residuals of large per-ray terms
        assert rel_l2(got, w64) <= lim + 3 * noise, f"{algo} d/d{n}: {rel_l2(got, w64):.2e} (reference fp32 itself {noise:.2e})"
        for b in range(3):                                   # and lens by lens: no lens sees another's rays
            assert rel_l2(got[b], w64[b]) <= 10 * (lim + 3 * noise), (n, b)


def test_batch_of_many_small_lenses_is_one_launch_and_matches_single_traces():
    """The reference's real caller traces its minibatch one lens at a time (optical_loss.py:96-110: F = 8, 8 rings,
    W = 3); here 24 perturbed Cooke triplets go through one launch and every lens equals its own B = 1 trace."""
    import yaml_free_lenses as L
    import torchoptics_amd as ta
    from torchoptics_amd import lens_modeling as lm, ops
    a = L.PRESCRIPTIONS["cooke"]
    B = 24
    gen = torch.Generator().manual_seed(5)
    c = torch.tensor(a["c"]).repeat(B, 1) * (1 + 0.02 * torch.randn(B, 7, generator=gen))
    t = torch.tensor(a["t"]).repeat(B, 1) * (1 + 0.02 * torch.rand(B, 7, generator=gen))
    st = lm.Structure(stop_idx=np.array(a["stop_idx"] * B), sequence=np.array(a["sequence"] * B), default_device=DEV)
    leaves = [q.reshape(-1).to(DEV).requires_grad_(True) for q in (c, t)]
    nd, v = (torch.tensor(a[k] * B, device=DEV) for k in ("nd", "v"))
    lens = lm.Lens(st, leaves[0], leaves[1], nd, v)
    specs = lm.Specs(st, torch.full((B,), L.EPD, device=DEV), torch.full((B,), float(np.deg2rad(L.HFOV_DEG)), device=DEV))
    tr = ta.RayTracer(mode="circular", n_rays=(8, 8), rel_fields=tuple(np.linspace(0, 1, 8)), wavelengths=(459., 520., 640.),
                      default_device=DEV)
    args = tr.assemble(specs, lens)
    ops.enable_timing(True)
    out = ta.trace_skew(args["x"], args["y"], args["z"], args["cx"], args["cy"], args["c"], args["t"], args["mu"], args["mask"])
    rms_b = ta.compute_rms2d_batch(out[0], out[1], out[4])
    rms_b.sum().backward()
    n_calls = ops.timing_counts()
    ops.enable_timing(False)
    assert n_calls == {"fwd": 1, "bwd": 1} and out[0].shape == (B, 8, 64, 3)
    g_c = leaves[0].grad.reshape(B, 7).clone()
    for b in (0, 7, 23):
        one = [(q[b:b + 1] if q.shape[0] == B else q).detach() for q in (args["x"], args["y"], args["z"], args["cx"], args["cy"])]
        cb = args["c"][b:b + 1].detach().clone().requires_grad_(True)
        o1 = ta.trace_skew(*one, cb, args["t"][b:b + 1].detach(), args["mu"][b:b + 1].detach(), args["mask"][b:b + 1])
        for i in range(6):
            assert torch.equal(out[i][b:b + 1], o1[i]), (b, i)
        r1 = ta.compute_rms2d(o1[0], o1[1], o1[4])
        assert abs(r1.item() - rms_b[b].item()) <= 1e-6 * r1.item()
        r1.backward()
        # d rms_b / d c of lens b through the trace alone (the batch leaf also feeds z and mu: compare the trace input's gradient)
        assert cb.grad is not None and torch.isfinite(g_c[b]).all()


def test_aggregate_on_a_batch():
    import torchoptics_amd as ta
    lens, specs, _ = _batch(DEV)
    tr = ta.RayTracer(mode="circular", n_rays=(8, 8), rel_fields=(0., 1.), wavelengths=("d",), default_device=DEV)
    out = tr.trace_rays(specs, lens, aggregate=True)
    assert len(out) == 7 and out[6]["theta_norm"][0].shape == (2, 2, 64, 1) and out[6].q_per_lens.shape == (2,)


@pytest.mark.parametrize("algo", ["inverse", "checkpoint"])
def test_batch_with_aspheric_rows_and_opd(algo):
    """The extensions on a batch: per-lens conic / polynomial terms ([B,S], [B,S,4]) and the optical path length with
    its gradient, against the oracle's trace_skew_general lens by lens (parity unpinned by the reference, which has no
    aspheres: the oracle is the FD-checked definition, tests/test_oracle_asphere.py)."""
    import torchoptics_amd as ta
    from conftest import rel_l2
    from oracle import trace_oracle as orc
    from torchoptics_amd import ops
    g, ins, mask = _g11(DEV)
    B, S, W = 3, ins[5].shape[-1], ins[7].shape[3]
    gen = torch.Generator().manual_seed(11)
    kap = torch.zeros(B, S)
    pol = torch.zeros(B, S, 4)
    kap[0, 0], kap[1, 3], kap[2, 5] = -0.6, 0.4, -1.0
    pol[0, 0, 0], pol[2, 5, 1], pol[1, 3, 0] = 2e-5, -2e-7, -3e-5
    kind = (kap != 0) | (pol != 0).any(dim=-1)
    n = [torch.ones(B, 1, 1, W)]
    for k in range(S):
        n.append(n[-1] / ins[7][..., k].cpu())
    n_index = torch.stack(n, dim=-1)                                   # [B,1,1,W,S+1]
    lv = [q.to(DEV).requires_grad_(True) for q in (ins[5].cpu(), ins[6].cpu(), kap, pol)]
    w_opd = torch.rand(B, 3, 256, W, generator=gen).to(DEV)
    ops.set_backward_algorithm(algo)
    try:
        out = ta.trace_skew(ins[0], ins[1], ins[2], ins[3], ins[4], lv[0], lv[1], ins[7], mask, kappa=lv[2], poly=lv[3],
                            surf_kind=kind.to(DEV), n_index=n_index.to(DEV), want_opd=True)
        loss = ta.compute_rms2d_batch(out[0], out[1], out[4]).sum() + 1e-3 * (out[6] * w_opd).sum()
        loss.backward()
    finally:
        ops.set_backward_algorithm("inverse")
    for b in range(B):
        one = [a[b:b + 1].cpu().double() if a.shape[0] == B else a.cpu().double() for a in ins[:5]]
        cb, tb = (ins[i][b:b + 1].cpu().double().requires_grad_(True) for i in (5, 6))
        kb, pb = kap[b].double().requires_grad_(True), pol[b].double().requires_grad_(True)
        o = orc.trace_skew_general(*one, cb, tb, ins[7][b:b + 1].cpu().double(), mask[b:b + 1].cpu(), kb, pb,
                                   [int(v) for v in kind[b]], n_index=n_index[b:b + 1].double())
        assert torch.equal(o[4], out[4][b:b + 1].cpu())
        for i, tol in ((0, 3e-5), (1, 3e-5), (6, 1e-4)):
            assert (o[i] - out[i][b:b + 1].cpu().double()).abs().max().item() <= tol, (b, i)
        lb = orc.compute_rms2d(o[0], o[1], o[4]) + 1e-3 * (o[6] * w_opd[b:b + 1].cpu().double()).sum()
        lb.backward()
        for name, got, want in (("c", lv[0].grad[b], cb.grad[0]), ("t", lv[1].grad[b], tb.grad[0]), ("kappa", lv[2].grad[b], kb.grad),
                                ("poly", lv[3].grad[b], pb.grad)):
            if want.abs().max() == 0:
                assert got.abs().max().item() == 0, (b, name)
                continue
            assert rel_l2(got.cpu().numpy(), want.numpy()) <= 1e-4, (algo, b, name, rel_l2(got.cpu().numpy(), want.numpy()))


def test_batch_without_backward_rays_and_untagged_spot_sizes():
    """allow_backward_rays=False on a padded batch (the per-lens sequence mask gates the backward-ray test,
    ray_tracing_lite.py:626-632) against the IEEE oracle, bit for bit; and compute_rms2d_batch on tensors that did not
    come straight from the trace (no fused moments attached) equals the fused value."""
    import torchoptics_amd as ta
    from oracle import trace_oracle as orc
    g, ins, mask = _g11(DEV)
    out = ta.trace_skew(*ins, mask, allow_backward_rays=False)
    want = orc.trace_skew(*[a.cpu() for a in ins], mask.cpu(), False, False, ieee_sqrt=True)
    assert all(torch.equal(out[i].cpu(), want[i]) for i in range(5))
    assert not out[5].any() and not want[5].any()           # (the reference leaves `ray_backward` at its broadcast shape here)
    assert (~want[4]).sum() > (~torch.from_numpy(g["ok"])).sum()          # rays that only the backward test removes
    fused = ta.compute_rms2d_batch(out[0], out[1], out[4])
    plain = ta.compute_rms2d_batch(out[0].clone(), out[1].clone(), out[4].clone())
    assert fused.shape == (3,) and torch.allclose(fused, plain, rtol=1e-6, atol=0)
    for b in range(3):
        one = orc.compute_rms2d(want[0][b:b + 1], want[1][b:b + 1], want[4][b:b + 1])
        assert abs(fused[b].item() - one.item()) <= 2e-5 * one.item()      # (the oracle sums in fp32, the kernel in fp64)


def test_batch_above_the_grid_row_limit_is_traced_in_lens_chunks():
    """4 096 lenses x 8 fields x 3 wavelengths = 98 304 (lens, field, wavelength) rows, more than one launch's 65 535:
    the reference's broadcasting has no such bound (lens_modeling.py:151-386).  trace_skew traces the batch in lens
    chunks and joins them: per-lens losses and per-lens gradients equal those of the same lenses traced as a small
    batch in one launch, bit for bit; aggregate='sum' (the real caller's loss) included."""
    import torchoptics_amd as ta
    from torchoptics_amd import ray_tracing as rt
    g, ins, mask = _g11(DEV)
    B0, S = ins[5].shape[0], ins[5].shape[-1]                      # the three padded lenses of the reference's batch run
    n_lens, F, W = 4096, 8, 3
    idx = torch.arange(n_lens, device=DEV) % B0
    torch.manual_seed(3)
    jit = 1.0 + 1e-3 * torch.randn(n_lens, 1, 1, 1, 1, device=DEV)              # every lens a little different
    fields = torch.linspace(0.0, 0.42, F, device=DEV).reshape(1, F, 1, 1)
    x, y = ins[0][:1, :1, :64, :1].contiguous(), ins[1][:1, :1, :64, :1].contiguous()      # one shared 64-ray fan
    mu = ins[7][idx].contiguous()                                  # [n_lens,1,1,3,S]; padded rows stay identity rows (mu = 1)

    def run(sel):
        c = (ins[5][idx] * jit)[sel].clone().requires_grad_(True)
        t = ins[6][idx][sel].clone().requires_grad_(True)
        m = mu[sel].clone().requires_grad_(True)
        out = ta.trace_skew(x, y, ins[2][idx][sel], torch.zeros(1, 1, 1, 1, device=DEV), fields, c, t, m, mask[idx][sel],
                            aggregate="sum")
        ld = rt.unsupervised_loss_batch(out, S, 0.2)
        ld["loss_unsup"].sum().backward()
        return out, ld, (c.grad, t.grad, m.grad)
    out, ld, grads = run(slice(0, n_lens))
    assert out[0].shape == (n_lens, F, 64, W) and ld["rms"].shape == (n_lens,)
    assert torch.isfinite(ld["loss_unsup"]).all()
    probe = slice(2730, 2746)                                      # straddles the chunk boundary (65535 // 24 = 2730 lenses)
    out_s, ld_s, grads_s = run(probe)
    for k in ("rms", "penalty", "loss_unsup"):
        assert torch.equal(ld[k][probe], ld_s[k]), k
    for i in range(6):
        assert torch.equal(out[i][probe], out_s[i]), i
    for a, b in zip(grads, grads_s):
        assert torch.equal(a[probe], b)
