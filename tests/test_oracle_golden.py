"""
Pins the CPU oracle (oracle/trace_oracle.py) to the reference's own outputs.

The fixtures were produced by tests/golden/make_golden.py, which ran the reference
(torchlens.ray_tracing_lite) on CPU.  Forward must be BIT-EXACT in fp32; gradients
w.r.t. the trace inputs must equal the reference autograd's.
"""
import numpy as np
import pytest
import torch

from conftest import assert_matches_fixture, load_golden, rel_l2, same_cpu_math
from oracle import trace_oracle as orc

RAY_CASES = ["G1_singlet_cfg1", "G2_cooke_16x16", "G4_doublet_32x32", "G4_tessar_32x32",
             "G5_cooke_failures", "G6_cooke_aim1", "G10_cooke_noback", "G10_tessar_noback"]


def _inputs(g, dtype=torch.float32, grad=False):
    names = ("in_x", "in_y", "in_z", "in_cx", "in_cy", "in_c", "in_t", "in_mu")
    ins = [torch.from_numpy(g[n]).to(dtype).requires_grad_(grad) for n in names]
    return ins, torch.from_numpy(g["in_mask"]), bool(g.get("allow_backward_rays", True))


@pytest.mark.parametrize("case", RAY_CASES)
def test_forward_bit_exact(case):
    g = load_golden(case)
    ins, mask, allow = _inputs(g)
    x, y, cx, cy, ok, back = orc.trace_skew(*ins, mask, False, allow)
    # (tolerances only apply on a CPU whose torch kernels round differently from the fixture machine's:
    #  grazing rays of the failure-heavy fan amplify a 1-ulp sqrt difference to ~3e-5 mm)
    scale = 10 if "failures" in case else 1          # rays within rounding of a failure threshold
    for name, got, tol in (("x", x, 5e-5), ("y", y, 5e-5), ("cx", cx, 2e-6), ("cy", cy, 2e-6)):
        assert_matches_fixture(got.numpy(), g[name], atol=tol * scale, what=f"{case}:{name}")
    assert np.array_equal(ok.numpy(), g["ok"])
    assert np.array_equal(back.numpy(), g["back"])
    rms = orc.compute_rms2d(x, y, ok)
    assert_matches_fixture(np.float32(rms.item()), np.float32(g["rms_in"]), atol=0, rtol=5e-6, what=f"{case}:rms")
    # failed rays come out as exact zeros
    dead = ~g["ok"] if allow else ~ok.numpy()
    if allow:
        assert not np.any(x.numpy()[dead]) and not np.any(y.numpy()[dead])


@pytest.mark.parametrize("case", RAY_CASES)
@pytest.mark.parametrize("prec", ["", "64"])
def test_input_gradients_equal_reference(case, prec):
    g = load_golden(case)
    dtype = torch.float64 if prec else torch.float32
    ins, mask, allow = _inputs(g, dtype, grad=True)
    x, y, cx, cy, ok, back = orc.trace_skew(*ins, mask, False, allow)
    rms = orc.compute_rms2d(x, y, ok)
    gs = torch.autograd.grad(rms, ins, allow_unused=True)
    assert abs(float(rms.detach()) - float(g["rms_in" + prec])) <= (0 if same_cpu_math() else 5e-6) * float(g["rms_in" + prec])
    for n, got in zip(("x", "y", "z", "cx", "cy", "c", "t", "mu"), gs):
        want = g["gin_" + n + prec]
        got = np.zeros_like(want) if got is None else got.numpy()
        if same_cpu_math():
            assert np.array_equal(got, want), f"{case}: d/d{n} differs (rel {rel_l2(got, want):.2e})"
        elif np.linalg.norm(g["gin_" + n + "64"]) > 1e-6 * np.linalg.norm(g["gin_c64"]):
            # another CPU: the same autograd graph, rounded differently (fp32 noise floor of this case)
            noise = rel_l2(g["gin_" + n], g["gin_" + n + "64"])
            lim = 1e-9 if prec else (3 * noise + 1e-5 if n in ("c", "t", "mu") else 1e-3)
            assert rel_l2(got, want) <= lim, f"{case}: d/d{n} rel {rel_l2(got, want):.2e}"


@pytest.mark.parametrize("case", ["G3_cooke_cfg2_d", "G3_cooke_cfg2_CdF"])
def test_cfg2_scalars_present(case):
    g = load_golden(case)
    assert g["in_c"].shape[-1] == 7 and np.all(g["ok_frac"] == 1.0)
    assert abs(float(g["rms_in"]) - float(g["rms_in64"])) < 1e-6


def test_moment_closed_form_matches_rms():
    """rms_from_moments (used by the sharded path) == compute_rms2d, failure-heavy case."""
    g = load_golden("G5_cooke_failures")
    ins, mask, allow = _inputs(g, torch.float64)
    x, y, cx, cy, ok, back = orc.trace_skew(*ins, mask, False, allow)
    want = orc.compute_rms2d(x, y, ok)
    m = orc.spot_moments(y, ok)
    got = orc.rms_from_moments(m, y.shape[2] * y.shape[3])
    assert abs(float(got) - float(want)) < 1e-13
    assert 0.5 < float(ok.double().mean()) < 0.9


def test_aggregate_stacks_and_penalty():
    g = load_golden("G7_harness_cooke")
    ins, mask, allow = _inputs(g)
    out = orc.trace_skew(*ins, mask, True, True)
    stacks = out[6]
    for key in ("z_RELU", "theta_norm", "theta_prime_norm"):
        got = torch.stack(stacks[key], 0).numpy()
        assert_matches_fixture(got, g["stack_" + key], atol=2e-4 if key != "z_RELU" else 1e-5, what=key)
    pen = orc.penalty_from_stacks(stacks, int(g["n_sequence"]))
    assert_matches_fixture(np.float32(pen.item()), np.float32(g["penalty"]), atol=0, rtol=2e-6, what="penalty")
    rms = orc.compute_rms2d(out[0], out[1], out[4])
    assert_matches_fixture(np.float32(rms.item()), np.float32(g["rms"]), atol=0, rtol=5e-6, what="rms")


@pytest.mark.parametrize("case", RAY_CASES)
def test_ieee_sqrt_variant_is_within_ulps_of_reference(case):
    """The correctly rounded variant (what the strict kernels match bit for bit) vs the
    reference, whose MKL sqrt is up to 1 ulp off per call: same masks, positions within 1e-5 mm
    (<= 10 ulp at |y| ~ 8 mm; measured worst 7.2e-6 on the failure-heavy fan), cosines within 5e-7."""
    g = load_golden(case)
    ins, mask, allow = _inputs(g)
    x, y, cx, cy, ok, back = orc.trace_skew(*ins, mask, False, allow, ieee_sqrt=True)
    assert np.array_equal(ok.numpy(), g["ok"]) and np.array_equal(back.numpy(), g["back"])
    assert np.abs(x.numpy() - g["x"]).max() <= 1e-5 and np.abs(y.numpy() - g["y"]).max() <= 1e-5
    assert np.abs(cx.numpy() - g["cx"]).max() <= 5e-7 and np.abs(cy.numpy() - g["cy"]).max() <= 5e-7
    rms = orc.compute_rms2d(x, y, ok)
    assert abs(float(rms) - float(g["rms_in"])) <= 1e-6 * float(g["rms_in"]) + 1e-9


def _batch_inputs(g, dtype=torch.float32, grad=False):
    names = ("in_x", "in_y", "in_z", "in_cx", "in_cy", "in_c", "in_t", "in_mu")
    return [torch.from_numpy(g[n]).to(dtype).requires_grad_(grad) for n in names], torch.from_numpy(g["in_mask"])


def test_padded_lens_batch_forward_equals_reference():
    """G11: three padded lenses (7, 5, 8 rows) in ONE trace_skew call of the reference; padded rows are identity rows."""
    g = load_golden("G11_batch3_16x16")
    ins, mask = _batch_inputs(g)
    assert ins[5].shape == (3, 1, 1, 1, 8) and not g["in_mask"][1, 0, 0, 0, 5:].any()
    out = orc.trace_skew(*ins, mask)
    for name, got, tol in zip(("x", "y", "cx", "cy"), out[:4], (5e-5, 5e-5, 2e-6, 2e-6)):
        assert_matches_fixture(got.numpy(), g[name], atol=tol, what=f"G11:{name}")
    assert np.array_equal(out[4].numpy(), g["ok"]) and np.array_equal(out[5].numpy(), g["back"])
    full = [ins[0].expand(3, 3, 256, 3), ins[1].expand(3, 3, 256, 3)] + ins[2:]
    stacks = orc.trace_skew(*full, mask, True, True)[6]
    for key in ("z_RELU", "theta_norm", "theta_prime_norm"):
        assert_matches_fixture(torch.stack(stacks[key], 0).numpy(), g["stack_" + key],
                               atol=2e-4 if key != "z_RELU" else 1e-5, what="G11:" + key)


@pytest.mark.parametrize("prec", ["", "64"])
def test_padded_lens_batch_gradients_equal_reference(prec):
    g = load_golden("G11_batch3_16x16")
    dtype = torch.float64 if prec else torch.float32
    ins, mask = _batch_inputs(g, dtype, grad=True)
    x, y, _, _, ok, _ = orc.trace_skew(*ins, mask)
    rms_b = torch.stack([orc.compute_rms2d(x[b:b + 1], y[b:b + 1], ok[b:b + 1]) for b in range(3)])
    assert np.allclose(rms_b.detach().numpy(), g["rms_b" + prec], rtol=0 if same_cpu_math() else 5e-6, atol=0)
    gs = torch.autograd.grad(rms_b.sum(), ins, allow_unused=True)
    for n, got in zip(("x", "y", "z", "cx", "cy", "c", "t", "mu"), gs):
        if n in ("x", "y"):
            continue
        want = g["gin_" + n + prec]
        got = np.zeros_like(want) if got is None else got.numpy()
        if same_cpu_math():
            assert np.array_equal(got, want), f"G11: d/d{n} differs (rel {rel_l2(got, want):.2e})"
        else:
            assert rel_l2(got, want) <= (1e-9 if prec else 1e-3), f"G11: d/d{n} rel {rel_l2(got, want):.2e}"
