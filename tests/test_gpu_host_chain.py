"""The two host chains -- the C++ autograd functions of _tlx.so (default) and the Python ctypes wrappers of ops.py --
launch the same kernels with the same arguments: every output, every gradient and the choice of backward algorithm
must agree BIT FOR BIT, on every kind of call (broadcast argument shapes, lens batches, aspheric rows, penalty term,
optical path length, per-ray input gradients, saved-tensor hooks)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from test_oracle_asphere import asphere_params

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
IN = ("in_x", "in_y", "in_z", "in_cx", "in_cy", "in_c", "in_t", "in_mu")


@pytest.fixture(scope="module")
def ta():
    import torchoptics_amd
    from torchoptics_amd import _lib, ops
    _lib.lib()
    assert ops.host_chain() == "cpp", "the C++ host extension (_tlx.so) is not built / does not load"
    return torchoptics_amd


def _both(fn):
    """fn() under the C++ host chain and under the Python one; returns the two results."""
    from torchoptics_amd import ops
    res = []
    for chain in ("cpp", "python"):
        ops.set_host_chain(chain)
        try:
            assert ops.host_chain() == chain
            res.append(fn())
        finally:
            ops.set_host_chain("cpp")
    return res


def _same(a, b, what=""):
    assert len(a) == len(b), what
    for i, (p, q) in enumerate(zip(a, b)):
        if p is None or q is None:
            assert p is None and q is None, (what, i)
            continue
        assert p.shape == q.shape and p.dtype == q.dtype, (what, i, p.shape, q.shape)
        assert torch.equal(p, q), (what, i)


CASES = ["plain", "asph", "penalty", "penalty_stacks", "opd", "input_grads", "noback", "x_moments", "rms_and_moments"]


@pytest.mark.parametrize("case", CASES)
def test_cpp_and_python_host_chains_agree_bit_for_bit(ta, case):
    from torchoptics_amd import ops, ray_tracing as rt
    g = load_golden("G10_tessar_noback" if case == "noback" else "G4_tessar_32x32")
    S = g["in_c"].shape[-1]
    kap0, pol0, _ = asphere_params(S)

    def run():
        ins = [torch.from_numpy(g[n]).to(DEV) for n in IN]
        mask = torch.from_numpy(g["in_mask"]).to(DEV)
        lv = [ins[i].clone().requires_grad_(True) for i in (2, 4, 5, 6, 7)]            # z, cy, c, t, mu
        kw, extra_leaves = {}, []
        agg, allow = False, case != "noback"
        if case == "asph":
            k, p_ = kap0.to(DEV).requires_grad_(True), pol0.to(DEV).requires_grad_(True)
            kw.update(kappa=k, poly=p_)
            extra_leaves += [k, p_]
        if case == "penalty":
            agg = "sum"
        if case == "penalty_stacks":
            agg = True
        if case == "opd":
            mu = ins[7]
            n = [torch.ones(1, 1, 1, mu.shape[3], device=DEV)]
            for k_ in range(S):
                n.append(n[-1] / mu[..., k_])
            nidx = torch.stack(n, dim=-1).requires_grad_(True)
            kw.update(n_index=nidx, want_opd=True)
            extra_leaves.append(nidx)
        if case in ("x_moments", "rms_and_moments"):
            kw.update(x_moments=True)
        x_in, y_in = ins[0], ins[1]
        if case == "input_grads":
            F, P, W = ins[4].shape[1], ins[0].shape[2], ins[7].shape[3]
            x_in = ins[0].expand(1, F, P, W).clone().requires_grad_(True)
            y_in = ins[1].clone().requires_grad_(True)                                 # [1,1,P,1]: reduced over f, w
            extra_leaves += [x_in, y_in]
        out = ta.trace_skew(x_in, y_in, lv[0], ins[3], lv[1], lv[2], lv[3], lv[4], mask, agg, allow, **kw)
        loss = rt.compute_rms_spot_xy(out[0], out[1], out[4]) if case == "x_moments" else ta.compute_rms2d(out[0], out[1], out[4])
        if case == "rms_and_moments":
            # the spot metric fused into the C++ trace node AND a second loss on the moments themselves: their two
            # gradients meet inside that node's backward (autograd's accumulation under the Python chain)
            loss = loss + 0.5 * rt.compute_rms_spot_xy(out[0], out[1], out[4]) + 2.0 * ta.compute_rms2d(out[0], out[1], out[4])
        loss = loss + 1e-3 * (out[0] * out[2]).sum()                                    # dense seeds on x and cx as well
        if agg:
            loss = loss + 0.2 * rt.penalty_sum(out[6], S)
        if case == "opd":
            loss = loss + 1e-4 * out[6].sum()
        loss.backward()
        outs = list(out[:6]) + ([out[6]] if case == "opd" else [])
        if agg is True:
            outs += [torch.stack(out[6][k_], 0) for k_ in ("z_RELU", "theta_norm", "theta_prime_norm")]
        return outs, [q.grad for q in lv + extra_leaves], ops.used_walk_back(out[0]), loss.detach()
    (o1, g1, inv1, l1), (o2, g2, inv2, l2) = _both(run)
    assert inv1 == inv2
    assert torch.equal(l1, l2)
    _same(o1, o2, case + " outputs")
    _same(g1, g2, case + " gradients")
    assert all(q is not None and torch.isfinite(q).all() for q in g1)


def test_host_chains_agree_on_a_lens_batch_with_shared_and_per_lens_arguments(ta):
    """B = 3 padded lenses (the reference's own batch run, fixture G11): per-lens c, t, mu, z, cy; the fan x, y and cx
    shared by all lenses (gradients summed over the lens axis by the host chain)."""
    from torchoptics_amd import ray_tracing as rt
    g = load_golden("G11_batch3_16x16")

    def run():
        ins = [torch.from_numpy(g[n]).to(DEV) for n in IN]
        mask = torch.from_numpy(g["in_mask"]).to(DEV)
        x_sh = ins[0][:1].clone().requires_grad_(True)                 # one fan for the three lenses
        lv = [ins[i].clone().requires_grad_(True) for i in (2, 4, 5, 6, 7)]
        out = ta.trace_skew(x_sh, ins[1][:1], lv[0], ins[3], lv[1], lv[2], lv[3], lv[4], mask, "sum", True)
        ld = rt.unsupervised_loss_batch(out, 8, 0.2)
        ld["loss_unsup"].sum().backward()
        return list(out[:6]) + [ld["rms"], ld["penalty"]], [q.grad for q in lv + [x_sh]]
    (o1, g1), (o2, g2) = _both(run)
    _same(o1, o2, "batch outputs")
    _same(g1, g2, "batch gradients")
    assert g1[-1].shape == (1, 1, 256, 1)


def test_host_chains_agree_through_ray_tracer_with_ray_aiming_and_hooks(ta):
    """The whole RayTracer.trace_rays chain (ray aiming re-enters the tracer inside the forward and asks for per-ray input
    gradients), the real caller's loss, under saved-tensor hooks."""
    import yaml_free_lenses as L
    from torchoptics_amd import ray_tracing as rt

    def run():
        lens, specs, leaves = L.build("cooke", DEV, epd=8.578)
        tr = ta.RayTracer(mode="circular", n_rays=(16, 16), rel_fields=list(np.linspace(0, 1, 3)), wavelengths=[459., 520., 640.],
                          n_ray_aiming_iter=1, default_device=DEV)
        with torch.autograd.graph.save_on_cpu():
            out = tr.trace_rays(specs, lens, aggregate="sum")
            ld = rt.unsupervised_loss(out, 7, 0.2)
        ld["loss_unsup"].backward()
        return [ld[k].detach() for k in ("loss_unsup", "rms", "penalty")], [leaves[k].grad for k in ("c", "t", "nd", "v")]
    (o1, g1), (o2, g2) = _both(run)
    _same(o1, o2, "loss_dict")
    _same(g1, g2, "leaf gradients")


@pytest.mark.parametrize("n_seq_kind", ["int", "tensor"])
def test_fused_loss_dict_equals_the_op_sequence_bit_for_bit(ta, n_seq_kind):
    """unsupervised_loss[_batch] under the C++ host chain is ONE launch (tl_unsup_loss) and one autograd node; under the
    Python chain it is the op sequence rms + rate * (q / n_seq).to(float32).  Values of the three loss_dict entries and the
    leaf gradients must agree bit for bit -- also when 'rms' and 'penalty' are used next to 'loss_unsup' downstream."""
    from torchoptics_amd import ray_tracing as rt
    g = load_golden("G11_batch3_16x16")

    def run():
        ins = [torch.from_numpy(g[n]).to(DEV) for n in IN]
        mask = torch.from_numpy(g["in_mask"]).to(DEV)
        lv = [ins[i].clone().requires_grad_(True) for i in (2, 4, 5, 6, 7)]
        out = ta.trace_skew(ins[0], ins[1], lv[0], ins[3], lv[1], lv[2], lv[3], lv[4], mask, "sum", True)
        n_seq = 8 if n_seq_kind == "int" else torch.tensor([8., 7., 5.], dtype=torch.float64, device=DEV)
        ld = rt.unsupervised_loss_batch(out, n_seq, 0.2)
        w = torch.tensor([1.0, -0.5, 2.0], device=DEV)
        ((ld["loss_unsup"] * w).sum() + 0.3 * ld["rms"][1] - 1e-3 * ld["penalty"].sum()).backward()
        return [ld[k].detach() for k in ("loss_unsup", "rms", "penalty")], [q.grad for q in lv]
    (o1, g1), (o2, g2) = _both(run)
    _same(o1, o2, "loss_dict")
    _same(g1, g2, "gradients")
    assert all(torch.isfinite(q).all() for q in g1)


def test_fused_loss_dict_of_one_lens(ta):
    """B = 1: unsupervised_loss (0-dim entries, as the reference's compute_loss_out) -- fused under the C++ chain."""
    from torchoptics_amd import ray_tracing as rt
    g = load_golden("G4_tessar_32x32")

    def run():
        ins = [torch.from_numpy(g[n]).to(DEV) for n in IN]
        mask = torch.from_numpy(g["in_mask"]).to(DEV)
        lv = [ins[i].clone().requires_grad_(True) for i in (2, 4, 5, 6, 7)]
        out = ta.trace_skew(ins[0], ins[1], lv[0], ins[3], lv[1], lv[2], lv[3], lv[4], mask, "sum", True)
        ld = rt.unsupervised_loss(out, ins[5].shape[-1], 0.2)
        assert all(ld[k].dim() == 0 and ld[k].dtype == torch.float32 for k in ld)
        ld["loss_unsup"].backward()
        return [ld[k].detach() for k in ("loss_unsup", "rms", "penalty")], [q.grad for q in lv]
    (o1, g1), (o2, g2) = _both(run)
    _same(o1, o2, "loss_dict")
    _same(g1, g2, "gradients")
