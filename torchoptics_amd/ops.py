"""
autograd.Function wrappers around the C ABI (include/tl_trace.h).

`TraceFunction` stands where PyTorch autograd's recorded graph of the reference's
`trace_skew` stood (ray_tracing_lite.py:594-675): forward = one fused HIP kernel,
backward = one recompute-and-reverse HIP kernel.  `SpotMomentsFunction` is the reduction of
`compute_rms2d` (ray_tracing_lite.py:678-702) for tensors that did not come from the trace.

Per-ray buffers are allocated [B, F, W, P] (pupil index contiguous) and handed
out as [B, F, P, W] permuted views, which is the reference's logical shape.  A batch of B padded
lenses is ONE launch (tl_problem.B); B = 1 is the reference's callers' case.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import TL_NMOM, tl_problem

_MODES = {"strict": _lib.MODE_STRICT, "fast": _lib.MODE_FAST}
_default_mode = os.environ.get("TORCHOPTICS_AMD_MODE", "strict")
# backward algorithm: "checkpoint" = re-trace forwards keeping the per-surface states in registers;
# "inverse" = walk back from the forward kernel's outputs (tl_trace_bwd_from_outputs), used whenever the
# problem allows it (allow_backward_rays, no penalty term, per-ray outputs were produced)
_bwd_algo = os.environ.get("TORCHOPTICS_AMD_BWD", "inverse")


# host chain: "cpp" = the C++ autograd functions of _tlx.so (csrc/tl_torch.cpp: argument normalisation, allocation,
# the C-ABI calls and the autograd nodes in C++; the default when the extension is built), "python" = the ctypes
# wrappers below (same kernels, same numbers; ~2x the host time per step)
_host = os.environ.get("TORCHOPTICS_AMD_HOST", "cpp")
_ext_mod = None


def set_host_chain(name: str) -> None:
    global _host
    if name not in ("cpp", "python"):
        raise ValueError("host chain must be 'cpp' or 'python'")
    _host = name


def _ext():
    """The C++ host extension, or None (not built / another ABI / host chain 'python')."""
    global _ext_mod
    if _host != "cpp":
        return None
    if _ext_mod is None:
        try:
            _lib.lib()                                  # libtltrace.so first: clear error when THAT is missing
            from . import _tlx
            _ext_mod = _tlx if _tlx.abi_version() == _lib.TL_ABI_VERSION else False
        except ImportError:
            _ext_mod = False
    return _ext_mod or None


def host_chain() -> str:
    """Which host chain trace_skew / compute_rms2d use right now: 'cpp' or 'python'."""
    return "cpp" if _ext() is not None else "python"


def used_walk_back(t: torch.Tensor) -> bool:
    """Whether the backward of the trace that produced `t` (the x returned by trace_skew) is the walk-back from the
    forward's outputs (tl_trace_bwd_from_outputs) rather than the checkpoint algorithm."""
    return bool(getattr(t, "_tl_use_inv"))


_last_use_inv = False


def set_backward_algorithm(name: str) -> None:
    global _bwd_algo
    if name not in ("checkpoint", "inverse"):
        raise ValueError("backward algorithm must be 'checkpoint' or 'inverse'")
    _bwd_algo = name


def get_backward_algorithm() -> str:
    return _bwd_algo


def set_default_mode(mode: str) -> None:
    """'strict' (bit-faithful fp32 op order) or 'fast' (FMA contraction, hardware rcp/sqrt)."""
    global _default_mode
    if mode not in _MODES:
        raise ValueError(f"mode must be one of {tuple(_MODES)}")
    _default_mode = mode


def get_default_mode() -> str:
    return _default_mode


def _require_device(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"torchoptics_amd: `{name}` lives on {t.device}; the ray tracer runs only as HIP kernels on an "
            "AMD GPU (there is no CPU fallback).  Move the lens and rays to device='cuda'.")


def _stream_ptr(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_ws_cache = {}

# optional live timing of the C-ABI calls with events on the launch stream (bench.py)
_timing = None


def enable_timing(on: bool = True) -> None:
    """Record a (start, stop) event pair around every tl_trace_fwd / tl_trace_bwd call."""
    global _timing
    _timing = {"fwd": [], "bwd": []} if on else None
    if _ext() is not None:
        _ext().enable_timing(bool(on))


def timing_counts() -> dict:
    """C-ABI trace calls recorded since enable_timing(True): {'fwd': n, 'bwd': n} (either host chain)."""
    out = {k: len(v) for k, v in (_timing or {}).items()}
    if _ext() is not None and _timing is not None:
        f, b = _ext().timing_counts()
        out = {"fwd": out.get("fwd", 0) + f, "bwd": out.get("bwd", 0) + b}
    return out


def timing_ms() -> dict:
    """Mean GPU milliseconds per call since enable_timing(); synchronises the device."""
    torch.cuda.synchronize()
    out = {k: (sum(a.elapsed_time(b) for a, b in v) / len(v) if v else None) for k, v in (_timing or {}).items()}
    if _ext() is not None and _timing is not None:
        f, b = _ext().timing_ms()
        out = {"fwd": f if f >= 0 else out.get("fwd"), "bwd": b if b >= 0 else out.get("bwd")}
    return out


class _Timed:
    def __init__(self, key, dev):
        self.key, self.dev = key, dev

    def __enter__(self):
        if _timing is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record(torch.cuda.current_stream(self.dev))

    def __exit__(self, *exc):
        if _timing is not None:
            self.b.record(torch.cuda.current_stream(self.dev))
            _timing[self.key].append((self.a, self.b))


class _NoCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_CTX = _NoCtx()


def _on_device(dev):
    """`with torch.cuda.device(dev)` only when dev is not the current device already (the usual case: ~6 us saved)."""
    return _NO_CTX if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)


def _workspace(nbytes: int, device) -> torch.Tensor:
    """Per-(device, stream) scratch for the block partials; grows monotonically."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


# hit-point slots handed to the walk-back kernel per lens (tl_problem.asph_hits): lenses with more aspheric rows than
# this still get the right gradient, from the checkpoint kernel (decided on the device, no host sync on the row kinds)
ASPH_HIT_SLOTS = 4


def set_asph_hit_slots(n: int) -> None:
    """Hit-point slots per lens for the walk-back over aspheric rows (0.._lib.TL_MAX_HIT_SLOTS); 0 = no stored hits:
    Newton iteration on the reversed ray (trace_bwd_inv_kernel<true>)."""
    global ASPH_HIT_SLOTS
    if not 0 <= int(n) <= _lib.TL_MAX_HIT_SLOTS:
        raise ValueError(f"hit slots must be 0..{_lib.TL_MAX_HIT_SLOTS}")
    ASPH_HIT_SLOTS = int(n)


def _problem(x_e, y_e, z, cx, cy, c, t, mu, mask_u8, allow_back, mode, kappa=None, poly=None, kind_u8=None,
             n_index=None, aggregate=False, hits=None, moments_x=False, cond=None):
    B, F, P, W = x_e.shape
    p = tl_problem()
    p.F, p.P, p.W, p.S = F, P, W, c.shape[-1]
    p.B = B
    p.xs_b, p.ys_b = x_e.stride(0), y_e.stride(0)
    p.device = x_e.device.index
    p.mode = _MODES[mode]
    p.allow_backward = 1 if allow_back else 0
    p.aggregate = 1 if aggregate else 0
    p.x_in, p.y_in = x_e.data_ptr(), y_e.data_ptr()
    p.xs_f, p.xs_p, p.xs_w = x_e.stride(1), x_e.stride(2), x_e.stride(3)
    p.ys_f, p.ys_p, p.ys_w = y_e.stride(1), y_e.stride(2), y_e.stride(3)
    p.z, p.cx, p.cy = z.data_ptr(), cx.data_ptr(), cy.data_ptr()
    # cx, cy: [1|B, 1|F] contiguous (a 1-D [1|F] is read as one lens)
    cx, cy = (a.reshape(1, -1) if a.dim() == 1 else a for a in (cx, cy))
    p.cx_stride, p.cx_stride_b = (0 if cx.shape[1] == 1 else 1), (0 if cx.shape[0] == 1 else cx.shape[1])
    p.cy_stride, p.cy_stride_b = (0 if cy.shape[1] == 1 else 1), (0 if cy.shape[0] == 1 else cy.shape[1])
    p.c, p.t, p.mu, p.mask = c.data_ptr(), t.data_ptr(), mu.data_ptr(), mask_u8.data_ptr()
    asph = kind_u8 is not None
    p.kappa = kappa.data_ptr() if asph else None
    p.poly = poly.data_ptr() if asph else None
    p.surf_kind = kind_u8.data_ptr() if asph else None
    p.n_index = n_index.data_ptr() if n_index is not None else None
    p.asph_hits = hits.data_ptr() if hits is not None else None
    p.asph_hit_slots = hits.shape[0] if hits is not None else 0
    p.moments_x = 1 if moments_x else 0
    p.cond_flags = cond.data_ptr() if cond is not None else None
    return p


def _fwp(t):
    """[B,F,P,W] logical tensor -> memory laid out [B,F,W,P] contiguous (no copy if it already is)."""
    return t.permute(0, 1, 3, 2).contiguous()


class TraceFunction(torch.autograd.Function):
    """(x, y, cx, cy, ok, back, moments, opd) = trace(x_in, y_in, z, cx, cy, c, t, mu[, kappa, poly]).

    Shapes: x_in, y_in [B,F,P,W] (expanded views welcome), z [B], cx, cy [1|B, 1|F], c, t [B,S], mu [B,W,S],
    mask_u8 [B,S]; kappa [B,S], poly [B,S,4], kind_u8 [B,S] are None for an all-spherical lens (the reference's
    case); n_index [B,W,S+1] is only needed for the optical path length output (`want_opd`).  moments [B*F, TL_NMOM]."""

    @staticmethod
    def forward(ctx, x_e, y_e, z, cx, cy, c, t, mu, kappa, poly, mask_u8, kind_u8, n_index, allow_back, mode,
                want_rays, want_opd, aggregate, want_stacks, moments_x=False):
        for name, ten in (("x", x_e), ("y", y_e), ("z", z), ("cx", cx), ("cy", cy), ("c", c), ("t", t),
                          ("mu", mu), ("mask", mask_u8)):
            _require_device(ten, name)
        dev = x_e.device
        B, F, P, W = x_e.shape
        S = c.shape[-1]
        if S > _lib.TL_MAX_SURFACES:
            raise RuntimeError(f"lens has {S} rows; this build supports at most {_lib.TL_MAX_SURFACES}")
        lib = _lib.lib()
        # per-ray input gradients (ray aiming: a handful of rays) keep the checkpoint algorithm: extreme rays
        # amplify the reconstruction rounding of the walk-back to ~1e-4 in d/dx_in, d/dy_in
        # ... and so does the gradient through the optical path length (only the checkpoint kernel carries it)
        use_inv = (_bwd_algo == "inverse" and want_rays and allow_back and not want_opd
                   and not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]))
        # aspheric rows: the forward leaves their hit points for the walk-back (8 B per ray and slot actually used)
        hits = None
        if use_inv and kind_u8 is not None and ASPH_HIT_SLOTS > 0 and any(ctx.needs_input_grad[2:10]):
            hits = torch.empty((min(ASPH_HIT_SLOTS, S), 2, B, F, W, P), dtype=torch.float32, device=dev)
        # one byte per ray: the forward flags the ill-conditioned live rays, so that the backward can leave exactly those
        # to the checkpoint kernel and walk back the rest (without it, one such ray costs the whole launch the walk-back)
        cond = (torch.empty((B, F, W, P), dtype=torch.uint8, device=dev)
                if use_inv and any(ctx.needs_input_grad[2:10]) else None)
        prob = _problem(x_e, y_e, z, cx, cy, c, t, mu, mask_u8, allow_back, mode, kappa, poly, kind_u8, n_index, aggregate,
                        hits, moments_x, cond)
        nbytes = lib.tl_workspace_bytes(C.byref(prob))
        ws = _workspace(nbytes, dev)
        if want_rays:
            fp = [torch.empty((B, F, W, P), dtype=torch.float32, device=dev) for _ in range(4)]
            bp = [torch.empty((B, F, W, P), dtype=torch.uint8, device=dev) for _ in range(2)]
        else:
            fp, bp = [None] * 4, [None] * 2
        opd = torch.empty((B, F, W, P), dtype=torch.float32, device=dev) if want_opd else None
        stacks = torch.empty((3, S, B, F, W, P), dtype=torch.float32, device=dev) if (aggregate and want_stacks) else None
        moments = torch.empty((B * F, TL_NMOM), dtype=torch.float64, device=dev)
        with _on_device(dev), _Timed("fwd", dev):
            rc = lib.tl_trace_fwd(C.byref(prob), *[_lib.ptr(b) for b in fp], *[_lib.ptr(b) for b in bp],
                                  _lib.ptr(opd), _lib.ptr(stacks), _lib.ptr(moments), _lib.ptr(ws), ws.numel(),
                                  _stream_ptr(dev))
        _lib.check(rc, "tl_trace_fwd")
        global _last_use_inv
        _last_use_inv = use_inv
        fwd_out = (fp[0], fp[1], fp[2], fp[3], bp[0], moments) if use_inv else (None,) * 6
        ctx.save_for_backward(x_e, y_e, z, cx, cy, c, t, mu, mask_u8, kappa, poly, kind_u8, *fwd_out,
                              n_index if want_opd else None, hits, cond)
        ctx.allow_back, ctx.mode, ctx.aggregate, ctx.use_inv = allow_back, mode, aggregate, use_inv
        ctx.prob, ctx.ws_bytes = prob, nbytes      # same tensors, same pointers in backward: no need to fill it again
        ctx.set_materialize_grads(False)
        if want_rays:
            outs = [b.permute(0, 1, 3, 2) for b in fp]
            flags = [b.view(torch.bool).permute(0, 1, 3, 2) for b in bp]
        else:
            outs = [torch.empty(0, device=dev) for _ in range(4)]
            flags = [torch.empty(0, dtype=torch.bool, device=dev) for _ in range(2)]
        opd_out = opd.permute(0, 1, 3, 2) if want_opd else torch.empty(0, device=dev)
        stk_out = stacks.permute(0, 1, 2, 3, 5, 4) if stacks is not None else torch.empty(0, device=dev)
        ctx.mark_non_differentiable(*flags, stk_out)
        if not want_opd:
            ctx.mark_non_differentiable(opd_out)
        return (*outs, *flags, moments, opd_out, stk_out)

    @staticmethod
    def backward(ctx, gx, gy, gcx, gcy, _gok, _gback, gmom, gopd, _gstk):
        (x_e, y_e, z, cx, cy, c, t, mu, mask_u8, kappa, poly, kind_u8, fx, fy, fcx, fcy, fok, fmom,
         n_index, hits, cond) = ctx.saved_tensors
        dev = x_e.device
        B, F, P, W = x_e.shape
        S = c.shape[-1]
        n_in = 20
        if gopd is not None and (n_index is None or gopd.numel() == 0):
            gopd = None
        if gx is None and gy is None and gcx is None and gcy is None and gmom is None and gopd is None:
            return (None,) * n_in
        asph = kind_u8 is not None
        lib = _lib.lib()
        prob = ctx.prob
        # saved-tensor hooks (save_on_cpu, checkpointing) hand back tensors in other storage than the forward saw
        if (prob.x_in != (x_e.data_ptr() or None) or prob.c != c.data_ptr() or prob.mu != mu.data_ptr()
                or prob.asph_hits != (hits.data_ptr() if hits is not None else None)
                or prob.cond_flags != (cond.data_ptr() if cond is not None else None)):
            prob = _problem(x_e, y_e, z, cx, cy, c, t, mu, mask_u8, ctx.allow_back, ctx.mode, kappa, poly, kind_u8, None,
                            ctx.aggregate, hits, False, cond)
        prob.n_index = n_index.data_ptr() if gopd is not None else None
        ws = _workspace(ctx.ws_bytes, dev)

        def dense(g):
            if g is None or g.numel() == 0:
                return None
            return _fwp(g.to(torch.float32))
        gxd, gyd, gcxd, gcyd, gopdd = dense(gx), dense(gy), dense(gcx), dense(gcy), dense(gopd)
        gmd = None if gmom is None else gmom.to(torch.float64).contiguous()
        need_xin, need_yin = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gxin = torch.empty((B, F, W, P), dtype=torch.float32, device=dev) if need_xin else None
        gyin = torch.empty((B, F, W, P), dtype=torch.float32, device=dev) if need_yin else None
        # one fp32 tensor per parameter group, written by the reduction kernel (fp64 sums rounded once):
        # autograd can take them as the leaves' .grad without a cast or a clone
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)     # noqa: E731
        parts = [new(B, S), new(B, S), new(B, W, S), new(B), new(B, F), new(B, F)]
        g_kappa, g_poly = (new(B, S), new(B, S, 4)) if asph else (None, None)
        g_n = new(B, W, S + 1) if gopdd is not None else None
        with _on_device(dev), _Timed("bwd", dev):
            if ctx.use_inv:
                rc = lib.tl_trace_bwd_from_outputs(
                    C.byref(prob), _lib.ptr(gxd), _lib.ptr(gyd), _lib.ptr(gcxd), _lib.ptr(gcyd), _lib.ptr(gmd),
                    _lib.ptr(fx), _lib.ptr(fy), _lib.ptr(fcx), _lib.ptr(fcy), _lib.ptr(fok), _lib.ptr(fmom),
                    *[_lib.ptr(q) for q in parts[:6]], _lib.ptr(g_kappa), _lib.ptr(g_poly), _lib.ptr(gxin),
                    _lib.ptr(gyin), _lib.ptr(ws), ws.numel(), _stream_ptr(dev))
            else:
                rc = lib.tl_trace_bwd(C.byref(prob), _lib.ptr(gxd), _lib.ptr(gyd), _lib.ptr(gcxd), _lib.ptr(gcyd),
                                      _lib.ptr(gmd), _lib.ptr(gopdd), *[_lib.ptr(q) for q in parts[:6]], _lib.ptr(g_kappa),
                                      _lib.ptr(g_poly), _lib.ptr(g_n), _lib.ptr(gxin), _lib.ptr(gyin), _lib.ptr(ws),
                                      ws.numel(), _stream_ptr(dev))
        _lib.check(rc, "tl_trace_bwd")
        g_c, g_t, g_mu, g_z, g_cx, g_cy = parts
        need = ctx.needs_input_grad
        def fold(g, like):          # [B,F] -> the (possibly broadcast) shape [1|B, 1|F] of cx / cy
            if like.shape[0] == 1 and B > 1:
                g = g.sum(dim=0, keepdim=True)
            if like.shape[1] == 1 and F > 1:
                g = g.sum(dim=1, keepdim=True)
            return g
        g_cx = fold(g_cx, cx) if need[3] else None
        g_cy = fold(g_cy, cy) if need[4] else None
        return (gxin.permute(0, 1, 3, 2) if need_xin else None,
                gyin.permute(0, 1, 3, 2) if need_yin else None,
                g_z.reshape(z.shape) if need[2] else None, g_cx.reshape(cx.shape) if need[3] else None,
                g_cy.reshape(cy.shape) if need[4] else None,
                g_c.reshape(c.shape), g_t.reshape(t.shape), g_mu.reshape(mu.shape),
                g_kappa.reshape(kappa.shape) if asph else None, g_poly.reshape(poly.shape) if asph else None,
                None, None, g_n.reshape(n_index.shape) if g_n is not None else None, None, None, None, None, None, None,
                None)


class TraceFunctionF64(torch.autograd.Function):
    """The trace in double precision (RayTracer(double_precision=True); tl_trace_fwd_f64 / tl_trace_bwd_f64: generic,
    untuned fp64 kernels, checkpoint backward).  Same argument shapes as TraceFunction with float64 tensors; no penalty
    term, no optical path length.  Returns (x, y, cx, cy, ok, back, moments)."""

    @staticmethod
    def forward(ctx, x_e, y_e, z, cx, cy, c, t, mu, kappa, poly, mask_u8, kind_u8, allow_back, want_rays):
        for name, ten in (("x", x_e), ("y", y_e), ("z", z), ("cx", cx), ("cy", cy), ("c", c), ("t", t), ("mu", mu)):
            _require_device(ten, name)
            if ten.dtype != torch.float64:
                raise TypeError(f"double-precision trace: `{name}` is {ten.dtype}")
        dev = x_e.device
        B, F, P, W = x_e.shape
        lib = _lib.lib()
        prob = _problem(x_e, y_e, z, cx, cy, c, t, mu, mask_u8, allow_back, "strict", kappa, poly, kind_u8)
        nbytes = lib.tl_workspace_bytes_f64(C.byref(prob))
        ws = _workspace(nbytes, dev)
        new = lambda dt: torch.empty((B, F, W, P), dtype=dt, device=dev)      # noqa: E731
        fp = [new(torch.float64) for _ in range(4)] if want_rays else [None] * 4
        bp = [new(torch.uint8) for _ in range(2)] if want_rays else [None] * 2
        moments = torch.empty((B * F, TL_NMOM), dtype=torch.float64, device=dev)
        with _on_device(dev):
            rc = lib.tl_trace_fwd_f64(C.byref(prob), *[_lib.ptr(b) for b in fp], *[_lib.ptr(b) for b in bp], _lib.ptr(moments),
                                      _lib.ptr(ws), ws.numel(), _stream_ptr(dev))
        _lib.check(rc, "tl_trace_fwd_f64")
        ctx.save_for_backward(x_e, y_e, z, cx, cy, c, t, mu, mask_u8, kappa, poly, kind_u8)
        ctx.allow_back = allow_back
        ctx.set_materialize_grads(False)
        if want_rays:
            outs = [b.permute(0, 1, 3, 2) for b in fp]
            flags = [b.view(torch.bool).permute(0, 1, 3, 2) for b in bp]
        else:
            outs = [torch.empty(0, dtype=torch.float64, device=dev) for _ in range(4)]
            flags = [torch.empty(0, dtype=torch.bool, device=dev) for _ in range(2)]
        ctx.mark_non_differentiable(*flags)
        return (*outs, *flags, moments)

    @staticmethod
    def backward(ctx, gx, gy, gcx, gcy, _gok, _gback, gmom):
        x_e, y_e, z, cx, cy, c, t, mu, mask_u8, kappa, poly, kind_u8 = ctx.saved_tensors
        if gx is None and gy is None and gcx is None and gcy is None and gmom is None:
            return (None,) * 14
        dev = x_e.device
        B, F, P, W = x_e.shape
        S = c.shape[-1]
        asph = kind_u8 is not None
        lib = _lib.lib()
        prob = _problem(x_e, y_e, z, cx, cy, c, t, mu, mask_u8, ctx.allow_back, "strict", kappa, poly, kind_u8)
        ws = _workspace(lib.tl_workspace_bytes_f64(C.byref(prob)), dev)

        def dense(g):
            return None if g is None or g.numel() == 0 else _fwp(g.to(torch.float64))
        gxd, gyd, gcxd, gcyd = dense(gx), dense(gy), dense(gcx), dense(gcy)
        gmd = None if gmom is None else gmom.to(torch.float64).contiguous()
        need = ctx.needs_input_grad
        new = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)     # noqa: E731
        gxin = new(B, F, W, P) if need[0] else None
        gyin = new(B, F, W, P) if need[1] else None
        g_c, g_t, g_mu, g_z, g_cx, g_cy = new(B, S), new(B, S), new(B, W, S), new(B), new(B, F), new(B, F)
        g_kappa, g_poly = (new(B, S), new(B, S, 4)) if asph else (None, None)
        with _on_device(dev):
            rc = lib.tl_trace_bwd_f64(C.byref(prob), _lib.ptr(gxd), _lib.ptr(gyd), _lib.ptr(gcxd), _lib.ptr(gcyd), _lib.ptr(gmd),
                                      _lib.ptr(g_c), _lib.ptr(g_t), _lib.ptr(g_mu), _lib.ptr(g_z), _lib.ptr(g_cx), _lib.ptr(g_cy),
                                      _lib.ptr(g_kappa), _lib.ptr(g_poly), _lib.ptr(gxin), _lib.ptr(gyin), _lib.ptr(ws), ws.numel(),
                                      _stream_ptr(dev))
        _lib.check(rc, "tl_trace_bwd_f64")

        def fold(g, like):
            if like.shape[0] == 1 and B > 1:
                g = g.sum(dim=0, keepdim=True)
            if like.shape[1] == 1 and F > 1:
                g = g.sum(dim=1, keepdim=True)
            return g.reshape(like.shape)
        return (gxin.permute(0, 1, 3, 2) if need[0] else None, gyin.permute(0, 1, 3, 2) if need[1] else None,
                g_z.reshape(z.shape) if need[2] else None, fold(g_cx, cx) if need[3] else None, fold(g_cy, cy) if need[4] else None,
                g_c.reshape(c.shape), g_t.reshape(t.shape), g_mu.reshape(mu.shape),
                g_kappa.reshape(kappa.shape) if asph else None, g_poly.reshape(poly.shape) if asph else None,
                None, None, None, None)


class SpotRmsFunction(torch.autograd.Function):
    """rms = compute_rms2d on the [F, TL_NMOM] moments (closed form) with its derivative, one tiny
    kernel each way instead of the ~25 elementwise kernels of the eager formula and its autograd.
    n_lens > 1: moments [n_lens * F, TL_NMOM] of a lens batch -> rms [n_lens], one value per lens."""

    @staticmethod
    def forward(ctx, moments, n_per_field, n_lens=1):
        _require_device(moments, "moments")
        dev = moments.device
        m = moments.to(torch.float64).contiguous()
        rms = torch.empty(() if n_lens == 1 else (n_lens,), dtype=torch.float32, device=dev)
        dm = torch.empty_like(m)
        with _on_device(dev):
            rc = _lib.lib().tl_spot_rms(dev.index, n_lens, m.shape[0] // n_lens, float(n_per_field), _lib.ptr(m),
                                        _lib.ptr(rms), _lib.ptr(dm), _stream_ptr(dev))
        _lib.check(rc, "tl_spot_rms")
        ctx.save_for_backward(dm)
        ctx.n_lens = n_lens
        return rms

    @staticmethod
    def backward(ctx, g):
        (dm,) = ctx.saved_tensors
        if ctx.n_lens == 1:
            return dm * g, None, None          # [F,10] fp64 * 0-dim fp32 -> fp64 in one launch (no separate cast)
        return (dm.view(ctx.n_lens, -1, TL_NMOM) * g.view(-1, 1, 1)).view_as(dm), None, None


class SpotMomentsFunction(torch.autograd.Function):
    """moments[F, TL_NMOM] of arbitrary per-ray tensors x, y, ok shaped [1, F, P, W]."""

    @staticmethod
    def forward(ctx, x, y, ok):
        _require_device(y, "y")
        dev = y.device
        _, F, P, W = y.shape
        if y.dtype != torch.float32 or any(s == 0 for s in y.stride()[1:]):
            y = y.to(torch.float32).contiguous()
        xs = None
        if x is not None:
            xs = x.to(torch.float32)
            if xs.stride() != y.stride():
                xs = torch.empty_strided(y.shape, y.stride(), dtype=torch.float32, device=dev).copy_(xs)
        oks = ok.view(torch.uint8) if ok.dtype == torch.bool else ok.to(torch.uint8)
        if oks.stride() != y.stride():
            oks = torch.empty_strided(y.shape, y.stride(), dtype=torch.uint8, device=dev).copy_(oks)
        lib = _lib.lib()
        moments = torch.empty((F, TL_NMOM), dtype=torch.float64, device=dev)
        ws = _workspace(F * W * ((P + 255) // 256) * TL_NMOM * 8 + 256, dev)
        with _on_device(dev):
            rc = lib.tl_spot_moments(dev.index, F, P, W, _lib.ptr(xs), _lib.ptr(y), _lib.ptr(oks),
                                     y.stride(1), y.stride(2), y.stride(3), _lib.ptr(moments),
                                     _lib.ptr(ws), ws.numel(), _stream_ptr(dev))
        _lib.check(rc, "tl_spot_moments")
        ctx.save_for_backward(xs if xs is not None else y, y, oks)
        ctx.has_x = xs is not None
        return moments

    @staticmethod
    def backward(ctx, gmom):
        xs, y, oks = ctx.saved_tensors
        dev = y.device
        _, F, P, W = y.shape
        lib = _lib.lib()
        gm = gmom.to(torch.float64).contiguous()
        gy = torch.empty_strided(y.shape, y.stride(), dtype=torch.float32, device=dev)
        need_x = ctx.has_x and ctx.needs_input_grad[0]
        gx = torch.empty_strided(y.shape, y.stride(), dtype=torch.float32, device=dev) if need_x else None
        with _on_device(dev):
            rc = lib.tl_spot_seed(dev.index, F, P, W, _lib.ptr(xs) if ctx.has_x else None, _lib.ptr(y),
                                  _lib.ptr(oks), y.stride(1), y.stride(2), y.stride(3), _lib.ptr(gm),
                                  _lib.ptr(gx), _lib.ptr(gy), _stream_ptr(dev))
        _lib.check(rc, "tl_spot_seed")
        return gx, gy, None


class PupilPositionFunction(torch.autograd.Function):
    """z [B] = paraxial entrance-pupil position from the rows in front of the stop: c, t [B,K], n [B,K+1]
    (tl_pupil_position: one tiny kernel forward, one backward, one thread per lens; mode 'strict': the value is the
    reference's fp32 product tree bit for bit, 'fast': fp64 inside, rounded once; default ops.get_default_mode())."""

    @staticmethod
    def forward(ctx, c, t, n, mode=None):
        ctx.mode = _MODES[mode or _default_mode]
        for name, ten in (("c", c), ("t", t), ("n", n)):
            _require_device(ten, name)
        B, K = c.shape
        c, t, n = (a.detach().to(torch.float32).contiguous() for a in (c, t, n))
        if t.shape != (B, K) or n.shape != (B, K + 1):
            raise ValueError("pupil position: c, t must hold [B,K] rows and n [B,K+1] indices")
        z = torch.empty(B, dtype=torch.float32, device=c.device)
        with _on_device(c.device):
            rc = _lib.lib().tl_pupil_position(c.device.index, B, K, _lib.ptr(c), _lib.ptr(t), _lib.ptr(n), _lib.ptr(z),
                                              None, None, None, None, ctx.mode, _stream_ptr(c.device))
        _lib.check(rc, "tl_pupil_position")
        ctx.save_for_backward(c, t, n)
        return z

    @staticmethod
    def backward(ctx, g_z):
        c, t, n = ctx.saved_tensors
        B, K = c.shape
        g_z = g_z.to(torch.float32).reshape(B).contiguous()
        g_c, g_t, g_n = torch.empty_like(c), torch.empty_like(t), torch.empty_like(n)
        with _on_device(c.device):
            rc = _lib.lib().tl_pupil_position(c.device.index, B, K, _lib.ptr(c), _lib.ptr(t), _lib.ptr(n), None,
                                              _lib.ptr(g_z), _lib.ptr(g_c), _lib.ptr(g_t), _lib.ptr(g_n), ctx.mode,
                                              _stream_ptr(c.device))
        _lib.check(rc, "tl_pupil_position (backward)")
        return g_c, g_t, g_n, None


# ------------------------------------------------------------------------------------------ entry points of the host chain
def _dist_initialized() -> bool:
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def trace_cpp(ext, x, y, z, cx, cy, c, t, mu, mask, kappa, poly, kind, n_index, allow_back, mode, want_rays, want_opd,
              aggregate, want_stacks, moments_x):
    """The trace through the C++ host extension: the arguments exactly as trace_skew received them (their broadcast
    shapes are normalised in C++).  Returns (the nine outputs of TraceFunction, use_inv)."""
    flags = ((ext.ALLOW_BACK if allow_back else 0) | (ext.WANT_RAYS if want_rays else 0) | (ext.WANT_OPD if want_opd else 0)
             | (ext.AGGREGATE if aggregate else 0) | (ext.WANT_STACKS if want_stacks else 0)
             | (ext.MOMENTS_X if moments_x else 0) | (ext.INVERSE if _bwd_algo == "inverse" else 0)
             # the spot metric of the traced rays as an output of the same node (ray_tracing.compute_rms2d picks it up);
             # not when the pupil is sharded over ranks: the moments are summed across them first
             # (nor with the penalty term: those callers take the whole loss_dict from the moments, unsup_loss below)
             | (ext.FUSE_RMS if (want_rays and not aggregate and hasattr(ext, "FUSE_RMS") and not _dist_initialized()) else 0))
    out = ext.trace(x, y, z, cx, cy, c, t, mu, mask, kappa, poly, kind, n_index, flags, _MODES[mode], ASPH_HIT_SLOTS)
    return out, ext.last_use_inv()


def pupil_position(c, t, n, mode=None):
    """PupilPositionFunction, through the C++ host extension when it is there."""
    ext = _ext()
    if ext is not None and c.is_cuda and hasattr(ext, "pupil_position"):
        return ext.pupil_position(c, t, n, _MODES[mode or _default_mode])
    return PupilPositionFunction.apply(c, t, n, mode)


def unsup_loss(moments, n_per_field, n_lens, n_sequence, penalty_rate):
    """(loss_unsup, rms, penalty) per lens from the moments of an aggregate trace in one launch (tl_unsup_loss), or None
    when the C++ host chain is not active (the callers then compose the same values from tensor ops)."""
    ext = _ext()
    if ext is None or not moments.is_cuda or not hasattr(ext, "unsup_loss"):
        return None
    if isinstance(n_sequence, (int, float)):
        return ext.unsup_loss(moments, float(n_per_field), int(n_lens), None, float(n_sequence), float(penalty_rate))
    return ext.unsup_loss(moments, float(n_per_field), int(n_lens), torch.as_tensor(n_sequence), 0.0, float(penalty_rate))


def spot_rms(moments, n_per_field, n_lens=1):
    """compute_rms2d on the moments (SpotRmsFunction), through the C++ host extension when it is there."""
    ext = _ext()
    if ext is not None and moments.is_cuda:
        return ext.spot_rms(moments, float(n_per_field), int(n_lens))
    return SpotRmsFunction.apply(moments, n_per_field, n_lens)
