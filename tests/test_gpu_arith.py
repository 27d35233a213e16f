"""Strict mode's division and square root (tl_div, tl_sqrt of csrc/tl_kernels.inc: 8 instructions each instead of the
compiler's 11 and 9) must be the correctly rounded IEEE results on every operand the trace can hand them: compared bit
for bit with numpy's float32 division and sqrt (both correctly rounded) on 2^26 operands spread over the ranges that occur
(cos-like values in (1e-6, 2), lengths up to 1e3 mm, both signs for the dividend) plus adversarial ones (values next to
powers of two, perfect squares +- 1 ulp, quotients next to rounding boundaries)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(a, b):
    from torchoptics_amd import _lib
    ta, tb = torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)
    q, r = torch.empty_like(ta), torch.empty_like(ta)
    rc = _lib.lib().tl_selftest_arith(0, _lib.MODE_STRICT, _lib.ptr(ta), _lib.ptr(tb), ta.numel(), _lib.ptr(q), _lib.ptr(r),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "tl_selftest_arith")
    torch.cuda.synchronize()
    return q.cpu().numpy(), r.cpu().numpy()


def _operands(n, seed):
    rng = np.random.default_rng(seed)
    k = n // 4
    # denominators / radicands: cos-like sums, squared cosines, lengths
    b = np.concatenate([rng.uniform(1e-6, 2.0, k), np.exp(rng.uniform(np.log(1e-6), np.log(1e3), k)),
                        rng.uniform(0.5, 1.0, k) ** 2, 1.0 - np.exp(rng.uniform(np.log(1e-7), np.log(0.5), k))]).astype(np.float32)
    a = (np.exp(rng.uniform(np.log(1e-6), np.log(1e3), n)) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    return a, b


def test_strict_division_and_sqrt_are_correctly_rounded_on_random_operands():
    for seed in range(4):                                   # 4 x 2^24 operands
        a, b = _operands(1 << 24, seed)
        q, r = _run(a, b)
        assert np.array_equal(q, a / b), f"division differs on {(q != a / b).sum()} of {a.size} operands"
        assert np.array_equal(r, np.sqrt(b)), f"sqrt differs on {(r != np.sqrt(b)).sum()} of {b.size} operands"


def test_strict_division_and_sqrt_on_adversarial_operands():
    rng = np.random.default_rng(7)
    # perfect squares and their neighbours; values next to powers of two; radicands whose root is next to a boundary
    base = rng.uniform(1e-3, 30.0, 1 << 20).astype(np.float32)
    sq = (base * base).astype(np.float32)
    b = np.concatenate([sq, np.nextafter(sq, np.float32(0)), np.nextafter(sq, np.float32(1e9)),
                        np.nextafter(np.exp2(rng.integers(-18, 10, 1 << 18)).astype(np.float32), np.float32(0)),
                        np.exp2(rng.integers(-18, 10, 1 << 18)).astype(np.float32)]).astype(np.float32)
    # dividends that make the quotient land next to a rounding boundary: q0 * b rounded, and its neighbours
    q0 = rng.uniform(1e-3, 1e3, b.size).astype(np.float32)
    a0 = (q0 * b).astype(np.float32)
    for a in (a0, np.nextafter(a0, np.float32(0)), np.nextafter(a0, np.float32(1e9)), -a0):
        q, r = _run(a.astype(np.float32), b)
        assert np.array_equal(q, a.astype(np.float32) / b)
        assert np.array_equal(r, np.sqrt(b))
