"""
GPU tests of the penalty term (trace_skew(aggregate=True)): the per-surface
stacks, the fused sum and its gradient, against fixture G7 (the reference's own
RaytracedOptics.do_ray_tracing: loss_dict, stacks, leaf gradients) and against the oracle.

Tolerances: z_RELU 1e-5 mm; theta stacks 3e-6 (device acosf vs the CPU's, near cos = 1 the acos
amplifies 1 ulp of cos to ~3e-4 of theta in RELATIVE terms, so the bound is absolute on theta/(pi/2));
penalty sum 2e-6 relative.  Gradients: d theta/d cos^2 = -1/(2 h u sqrt(1-u^2)) reaches ~2000 towards
normal incidence while d cos^2/d(inputs) vanishes there by cancellation, so rays within ~0.002 rad of a
surface normal carry percent-level fp32 noise in ANY fp32 evaluation (the oracle's own fp32 autograd is
2e-5..1e-3 from its fp64).  Bounds vs the fp64 oracle: lens parameters c, t, mu 1e-4 + 2x the oracle's
fp32 noise; launch conditions z, cy (cancellation-heavy sums) 2e-3 + 2x.  Measured on the 9k-ray cases:
1e-6..5e-6 from the oracle's fp32 (IEEE sqrt) autograd in every group.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu


def _used_walk_back(t):
    from torchoptics_amd import ops
    return ops.used_walk_back(t)
DEV = "cuda:0"
IN = ("in_x", "in_y", "in_z", "in_cx", "in_cy", "in_c", "in_t", "in_mu")


@pytest.fixture(scope="module")
def ta():
    import torchoptics_amd
    from torchoptics_amd import _lib
    _lib.lib()
    return torchoptics_amd


def test_stacks_and_loss_dict_match_reference_fixture(ta):
    from torchoptics_amd import ray_tracing as rt
    g = load_golden("G7_harness_cooke")
    ins = [torch.from_numpy(g[n]).to(DEV) for n in IN]
    out = ta.trace_skew(*ins, torch.from_numpy(g["in_mask"]).to(DEV), True, True)
    assert len(out) == 7
    stacks = out[6]
    assert set(stacks) == {"z_RELU", "theta_norm", "theta_prime_norm"} and len(stacks["z_RELU"]) == 7
    for key, tol in (("z_RELU", 1e-5), ("theta_norm", 3e-6), ("theta_prime_norm", 3e-6)):
        got = torch.stack(stacks[key], 0).cpu().numpy()
        assert got.shape == g["stack_" + key].shape
        assert np.abs(got - g["stack_" + key]).max() <= tol, key
    n_seq = int(g["n_sequence"])
    ld = rt.unsupervised_loss(out, n_seq, 0.2)
    assert abs(ld["penalty"].item() - float(g["penalty"])) <= 2e-6 * float(g["penalty"])
    assert abs(ld["rms"].item() - float(g["rms"])) <= 5e-6 * float(g["rms"])
    assert abs(ld["loss_unsup"].item() - float(g["loss_unsup"])) <= 2e-6 * float(g["loss_unsup"])
    # the value-only path (plain dict of tensors) gives the same number
    plain = {k: v for k, v in stacks.items()}
    assert abs(rt.penalty_sum(plain, n_seq).item() - ld["penalty"].item()) <= 1e-5 * ld["penalty"].item()


@pytest.mark.parametrize("case", ["G7_harness_cooke", "G5_cooke_failures", "G10_cooke_noback"])
def test_penalty_gradient_matches_oracle(ta, case):
    """d(sumQ)/d(c, t, mu, z, cy): includes rays that die part-way (they keep gradient through the
    surfaces they passed alive) and parked rays."""
    from oracle import trace_oracle as orc
    from torchoptics_amd import ray_tracing as rt
    g = load_golden(case)
    mask = torch.from_numpy(g["in_mask"])
    allow = bool(g.get("allow_backward_rays", True))
    n_seq = 7
    res = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        ins = [torch.from_numpy(g[n]).to(dt) for n in IN]
        lv = [ins[i].clone().requires_grad_(True) for i in (2, 4, 5, 6, 7)]
        o = orc.trace_skew(ins[0], ins[1], lv[0], ins[3], lv[1], lv[2], lv[3], lv[4], mask, True, allow,
                           ieee_sqrt=(dt == torch.float32))
        pen = orc.penalty_from_stacks(o[6], n_seq)
        pen.backward()
        res[tag] = (pen.item(), [q.grad for q in lv])
    ins = [torch.from_numpy(g[n]).to(DEV) for n in IN]
    lv = [ins[i].clone().requires_grad_(True) for i in (2, 4, 5, 6, 7)]
    o = ta.trace_skew(ins[0], ins[1], lv[0], ins[3], lv[1], lv[2], lv[3], lv[4], mask.to(DEV), True, allow)
    pen = rt.penalty_sum(o[6], n_seq)
    pen.backward()
    # acos near cos = 1 makes the fp32 sum itself ~7e-6 off its fp64 value (the reference's fp32 too)
    assert abs(pen.item() - res["f64"][0]) <= 3e-6 * abs(res["f64"][0]) + 2 * abs(res["f32"][0] - res["f64"][0])
    for n, q, g32, g64 in zip(("z", "cy", "c", "t", "mu"), lv, *[res[k][1] for k in ("f32", "f64")]):
        e64, noise = rel_l2(q.grad.cpu().numpy(), g64.numpy()), rel_l2(g32.numpy(), g64.numpy())
        print(f"{case} penalty d/d{n}: vs fp64 {e64:.2e} (oracle fp32 itself {noise:.2e})")
        assert e64 <= (1e-4 if n in ("c", "t", "mu") else 2e-3) + 2 * noise, f"{case} d/d{n}: {e64:.2e}"


def test_full_chain_harness_equivalent_matches_reference(ta):
    """The reference's RaytracedOptics.do_ray_tracing case of fixture G7 through this package:
    8x8 circular pupil, 3 fields, (459,520,640) nm, one ray-aiming iteration, aggregate=True."""
    import yaml_free_lenses as L
    from torchoptics_amd import ray_tracing as rt
    g = load_golden("G7_harness_cooke")
    lens, specs, leaves = L.build("cooke", DEV, epd=8.578)
    tr = ta.RayTracer(mode="circular", n_rays=(8, 8), rel_fields=list(np.linspace(0, 1, 3)), wavelengths=[459., 520., 640.],
                      n_ray_aiming_iter=1, default_device=DEV)
    out = tr.trace_rays(specs, lens, aggregate=True)
    ld = rt.unsupervised_loss(out, 7, 0.2)
    for k in ("loss_unsup", "rms", "penalty"):
        assert abs(ld[k].item() - float(g[k])) <= 2e-5 * abs(float(g[k])), (k, ld[k].item(), float(g[k]))
    grads = torch.autograd.grad(ld["loss_unsup"], [leaves[k] for k in ("c", "t", "nd")])
    for k, got in zip(("c", "t", "nd"), grads):
        err = rel_l2(got.cpu().numpy(), g["g_loss_unsup_" + k])
        assert err <= 3e-4, f"d loss_unsup / d{k}: {err:.2e}"     # fp32 autograd through acos near 1 is itself ~1e-4 noisy


def test_full_loss_of_the_real_caller_on_the_two_asphere_double_gauss(ta):
    """BASELINE configs[2] as written through the real caller's loss (rms + penalty_rate * sumQ,
    optics_simulator_lite.py:430-450): round 1 refused aggregate=True on aspheric lenses."""
    from torchoptics_amd import prescriptions as P, ray_tracing as rt
    lens, specs, leaves = P.double_gauss(DEV, aspheres=True)
    tr = ta.RayTracer(mode="circular", n_rays=(16, 16), rel_fields=(0., 0.707, 1.), wavelengths=[459., 520., 640.],
                      n_ray_aiming_iter=1, default_device=DEV)
    out = tr.trace_rays(specs, lens, aggregate=True)
    ld = rt.unsupervised_loss(out, 11, 0.2)
    assert all(torch.isfinite(ld[k]).item() for k in ("loss_unsup", "rms", "penalty"))
    assert abs(ld["loss_unsup"].item() - (ld["rms"].item() + 0.2 * ld["penalty"].item())) <= 1e-5 * abs(ld["loss_unsup"].item())
    ld["loss_unsup"].backward()
    for k in ("c", "t", "nd", "kappa", "poly"):
        assert leaves[k].grad is not None and torch.isfinite(leaves[k].grad).all(), k
    assert leaves["kappa"].grad[1].abs().item() > 0 and leaves["kappa"].grad[10].abs().item() > 0


def test_aggregate_sum_equals_aggregate_without_the_stacks(ta):
    """aggregate='sum': the loss_dict and its gradients of aggregate=True, bit for bit, with no per-surface tensors
    written (the reference's caller only ever sums them, optics_simulator_lite.py:441-448)."""
    import yaml_free_lenses as L
    from torchoptics_amd import ray_tracing as rt
    res = {}
    for agg in (True, "sum"):
        lens, specs, leaves = L.build("cooke", DEV, epd=8.578)
        tr = ta.RayTracer(mode="circular", n_rays=(8, 8), rel_fields=list(np.linspace(0, 1, 3)), wavelengths=[459., 520., 640.],
                          n_ray_aiming_iter=1, default_device=DEV)
        out = tr.trace_rays(specs, lens, aggregate=agg)
        assert len(out) == 7 and (len(out[6]) == (3 if agg is True else 0))
        ld = rt.unsupervised_loss(out, 7, 0.2)
        grads = torch.autograd.grad(ld["loss_unsup"], [leaves[k] for k in ("c", "t", "nd")])
        res[agg] = ([ld[k].item() for k in ("loss_unsup", "rms", "penalty")], [g.clone() for g in grads])
    assert res[True][0] == res["sum"][0]
    assert all(torch.equal(a, b) for a, b in zip(res[True][1], res["sum"][1]))


@pytest.mark.parametrize("mode", ["strict", "fast"])
def test_penalty_walk_back_takes_the_live_rays_and_the_checkpoint_pass_the_dead_ones(ta, mode):
    """The real caller's loss rms + 0.2 sumQ on a fan where every 7th ray starts far outside the aperture: those rays die
    at rows 0..3 and keep the penalty gradient of the rows they passed alive, while no LIVE ray is ill-conditioned -- so
    the default backward is the walk-back kernel for the live rays + the checkpoint kernel over the dead ones, summed by
    the reduction.  Against the checkpoint algorithm for every ray and against the oracle's fp64 autograd."""
    from oracle import trace_oracle as orc
    from torchoptics_amd import ops, ray_tracing as rt
    g = load_golden("G4_tessar_32x32")
    ins = [torch.from_numpy(g[n]) for n in IN]
    mask = torch.from_numpy(g["in_mask"])
    F, P, W, S = ins[4].shape[1], ins[0].shape[2], ins[7].shape[3], ins[5].shape[-1]
    x_in, y_in = ins[0].expand(1, F, P, W).clone(), ins[1].expand(1, F, P, W).clone()
    x_in[:, :, ::7] *= 6.0
    y_in[:, :, ::7] *= 6.0
    names = ("z", "cy", "c", "t", "mu")
    res = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        lv = [ins[i].to(dt).clone().requires_grad_(True) for i in (2, 4, 5, 6, 7)]
        o = orc.trace_skew(x_in.to(dt), y_in.to(dt), lv[0], ins[3].to(dt), lv[1], lv[2], lv[3], lv[4], mask, True, True,
                           ieee_sqrt=(dt == torch.float32))
        (orc.compute_rms2d(o[0], o[1], o[4]) + 0.2 * orc.penalty_from_stacks(o[6], S)).backward()
        res[tag] = [q.grad for q in lv]
        ok_ref = o[4]
    assert 0.05 < (~ok_ref).float().mean().item() < 0.2          # the fan has dead rays ...
    got = {}
    for algo in ("inverse", "checkpoint"):
        ops.set_backward_algorithm(algo)
        try:
            lv = [ins[i].to(DEV).clone().requires_grad_(True) for i in (2, 4, 5, 6, 7)]
            o = ta.trace_skew(x_in.to(DEV), y_in.to(DEV), lv[0], ins[3].to(DEV), lv[1], lv[2], lv[3], lv[4], mask.to(DEV),
                              "sum", True, mode=mode)
            assert _used_walk_back(o[0]) is (algo == "inverse")
            assert o[1]._tl_spot[0][:, 9].sum().item() == 0      # ... and no ill-conditioned live ray: the walk-back runs
            (ta.compute_rms2d(o[0], o[1], o[4]) + 0.2 * rt.penalty_sum(o[6], S)).backward()
            got[algo] = [q.grad.cpu() for q in lv]
        finally:
            ops.set_backward_algorithm("inverse")
    tol = 1e-4 if mode == "strict" else 5e-4
    for n, a, b, g32, g64 in zip(names, got["inverse"], got["checkpoint"], res["f32"], res["f64"]):
        e_ck, e64, noise = rel_l2(a.numpy(), b.numpy()), rel_l2(a.numpy(), g64.numpy()), rel_l2(g32.numpy(), g64.numpy())
        print(f"penalty walk-back {mode} d/d{n}: vs checkpoint {e_ck:.2e}, vs fp64 {e64:.2e} (oracle fp32 itself {noise:.2e})")
        assert not torch.equal(a, b)                             # two different algorithms did the work
        # (two fp32 evaluations of a gradient whose own fp32 noise is `noise`: near normal incidence d theta / d cos^2 is
        #  ~2000 and fp32 autograd itself is 1e-4 from fp64, see the module docstring)
        lim = tol if n in ("c", "t", "mu") else 2e-3
        assert e_ck <= lim + 2 * noise and e64 <= lim + 2 * noise, f"{mode} d/d{n}: {e_ck:.2e} / {e64:.2e}"


@pytest.mark.parametrize("log2p", [25, 20])
def test_selective_pass_finds_sparse_dead_rays_in_a_large_fan(ta, log2p):
    """The checkpoint pass behind the penalty walk-back reads the walk-back's scan map 64 chunks per look.  A fan of
    2^25 + 37 pupil points puts 128 chunks on every block of that launch (two looks per wave, a ragged last chunk); 2^20 + 37
    is the small-plan case.  Rays scattered over the fan (and the whole last, partial chunk) start outside the
    aperture and die (0.1 % of the fan, scattered: most chunks hold none): the result must equal the checkpoint algorithm
    on every ray up to the walk-back's rounding on the live ones; dropping the dead rays would show at ~1e-3."""
    from torchoptics_amd import ops, ray_tracing as rt
    g = load_golden("G4_tessar_32x32")
    ins = [torch.from_numpy(g[n]).to(DEV) for n in IN]
    mask = torch.from_numpy(g["in_mask"]).to(DEV)
    S = ins[5].shape[-1]
    P = (1 << log2p) + 37
    gen = torch.Generator(DEV).manual_seed(3)
    r = torch.sqrt(torch.rand(P, device=DEV, generator=gen)) * float(g["in_x"].max())
    th = torch.rand(P, device=DEV, generator=gen) * 6.2831853
    x, y = (r * torch.cos(th)), (r * torch.sin(th))
    dead = torch.randint(0, P, (P // 1000,), device=DEV, generator=gen)       # 0.1 % of the fan: visible at the 2e-5 gate below
    dead = torch.cat([dead, torch.arange(P - 37, P, device=DEV)])
    x[dead] *= 40.0
    y[dead] *= 40.0
    x, y = x.reshape(1, 1, P, 1), y.reshape(1, 1, P, 1)
    cy = ins[4][:, :1]                                           # one field, the fixture's three wavelengths
    F, W = 1, ins[7].shape[3]
    got = {}
    for algo in ("inverse", "checkpoint"):
        ops.set_backward_algorithm(algo)
        try:
            lv = [ins[i].clone().requires_grad_(True) for i in (5, 6, 7)]
            o = ta.trace_skew(x.expand(1, F, P, W), y.expand(1, F, P, W), ins[2], ins[3], cy, lv[0], lv[1], lv[2], mask, "sum", True)
            assert _used_walk_back(o[0]) is (algo == "inverse")
            n_dead = int((~o[4]).sum().item())
            (ta.compute_rms2d(o[0], o[1], o[4]) + 0.2 * rt.penalty_sum(o[6], S)).backward()
            got[algo] = [q.grad.clone() for q in lv]
            del o
        finally:
            ops.set_backward_algorithm("inverse")
    assert P // 1000 <= n_dead <= 3 * (P // 1000 + 37) + 64
    for n, a, b in zip(("c", "t", "mu"), got["inverse"], got["checkpoint"]):
        e = rel_l2(a.cpu().numpy(), b.cpu().numpy())
        print(f"2^{log2p} + 37 points, {n_dead} dead rays: d/d{n} walk-back + selective pass vs checkpoint {e:.2e}")
        assert e <= 2e-5, n
    torch.cuda.empty_cache()
