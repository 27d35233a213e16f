"""
Lens data model: drop-in for the reference's `torchlens/lens_modeling.py`.

Same public names, constructor signatures and tensor conventions as the reference
(`Structure` lens_modeling.py:151-213, `Specs` :216-252, `Lens` :255-386, `g_from_n_v` /
`n_v_from_g` :29-46), written from scratch.  These are the argument types of the hot path's
entry point `RayTracer.trace_rays`; everything here is O(rows) host-side tensor work and
stays in plain PyTorch (device-generic, dtype-generic) so gradients chain from the kernel's
d/d(mu, z, c, t) back to the leaves (c, t, nd, v) through ordinary autograd.

Tensors are 2-D padded [lens, row]: c and t padded with 0, nd with 1, v with NaN.  1-D
"flat" forms hold only the real rows (`Structure.mask`) or only the glass rows (`mask_G`).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

# Fraunhofer lines used by the two-term dispersion model [nm] (lens_modeling.py:362-364)
_LAMBDA_C, _LAMBDA_D, _LAMBDA_F = 656.3, 587.6, 486.1

# Small constant tensors are built once per (value, dtype, device): the per-step host chain must not
# issue host-to-device copies, otherwise it cannot be captured in a HIP graph (and each is a sync).
_CONST = {}


def const_tensor(values, dtype, device, shape=None):
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = (tuple(np.ravel(values).tolist()), dtype, device, shape)
    t = _CONST.get(key)
    if t is None:
        t = torch.tensor(values, dtype=dtype)
        if shape is not None:
            t = t.reshape(shape)
        t = _CONST[key] = t.to(device)
    return t


def mask_replace(mask: np.ndarray, src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """Write the 1-D `dst` into `src` where `mask` is set (row-major order)."""
    assert tuple(src.shape) == tuple(mask.shape), "mask and padded tensor must agree in shape"
    assert src.dtype == dst.dtype and src.device == dst.device and dst.dim() == 1
    return src.masked_scatter(torch.from_numpy(np.ascontiguousarray(mask)).to(dst.device), dst)


# glass variables: whitened (nd, v) pairs, constants from lens_modeling.py:33-34,43-44
_G_FROM_NV = [[-7.497527849096219, -7.49752916467739], [0.07842101471405442, -0.07842100095362642]]
_NV_FROM_G = [[-0.06668863644654068, 6.3758429552417315], [-0.0666886481483064, -6.375841836481304]]
_NV_MEAN = [[1.6426209211349487, 48.8505973815918]]


def g_from_n_v(n: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """(nd, v) -> normalised 2-vector glass variable g, shape [G, 2]."""
    assert n.dim() == 1 and v.dim() == 1 and n.device == v.device and n.dtype == v.dtype
    centred = torch.stack((n, v), dim=-1) - n.new_tensor(_NV_MEAN)
    return torch.matmul(centred, n.new_tensor(_G_FROM_NV))


def n_v_from_g(g: torch.Tensor):
    """Inverse of g_from_n_v: returns (nd, v)."""
    assert g.dim() == 2 and g.shape[1] == 2
    return torch.unbind(torch.matmul(g, g.new_tensor(_NV_FROM_G)) + g.new_tensor(_NV_MEAN), dim=1)


def map_glass_to_closest(g: torch.Tensor, catalog_g: torch.Tensor):
    """Nearest catalog glass (Euclidean in g space) for every row of g."""
    idx = torch.cdist(g[None], catalog_g[None])[0].argmin(dim=1) if g.numel() else g.new_zeros(0, dtype=torch.long)
    return catalog_g[idx], catalog_g


def find_valid_curvatures(sequence: "Structure") -> np.ndarray:
    """Rows whose curvature is a free variable (lens_modeling.py:49-53)."""
    prev_glass = np.concatenate((np.zeros_like(sequence.mask_G[:, :1]), sequence.mask_G[:, :-1]), axis=1)
    return sequence.mask_G | prev_glass & sequence.mask_except_last & sequence.mask


class Structure:
    """Which rows exist (`mask`), which are glass (`mask_G`), and where the stop is."""

    def __init__(self, stop_idx, mask: Optional[np.ndarray] = None, mask_G: Optional[np.ndarray] = None,
                 sequence=None, default_device='cuda'):
        self.stop_idx = np.asarray(stop_idx)
        assert self.stop_idx.ndim == 1, "stop_idx must be 1-dimensional (one entry per lens)"
        if sequence is not None:
            assert mask is None and mask_G is None, "give either `sequence` or the two masks"
            assert isinstance(sequence, np.ndarray)
            letters = np.asarray(sequence).astype(str)
            width = max((len(s) for s in letters), default=0)
            grid = np.array([list(s.ljust(width, '\0')) for s in letters]).reshape(len(letters), width)
            self.mask = grid != '\0'
            self.mask_G = grid == 'G'
        else:
            assert mask is not None and mask_G is not None
            self.mask, self.mask_G = np.asarray(mask, dtype=bool), np.asarray(mask_G, dtype=bool)
        assert self.mask.ndim == 2 and self.mask_G.ndim == 2
        self.default_device = default_device
        self.mask_torch = torch.from_numpy(np.ascontiguousarray(self.mask)).to(default_device)
        self.mask_G_torch = torch.from_numpy(np.ascontiguousarray(self.mask_G)).to(default_device)
        # flat positions of the real rows / the glass rows in the row-major padded layout: gathers and
        # scatters by index need no device->host synchronisation (boolean-mask indexing does), so the
        # per-step host chain stays capturable in a HIP graph
        self.idx_rows = torch.from_numpy(np.flatnonzero(self.mask)).to(default_device)
        self.idx_glass = torch.from_numpy(np.flatnonzero(self.mask_G)).to(default_device)

    def __len__(self):
        return self.mask.shape[0]

    def up_to_stop(self) -> "Structure":
        """The rows in front of the aperture stop (used to locate the entrance pupil).  Memoised: a
        Structure is immutable, and building one uploads its masks to the device."""
        cached = getattr(self, "_up_to_stop", None)
        if cached is None:
            width = int(self.stop_idx.max())
            before = np.arange(width)[None, :] < self.stop_idx[:, None]
            cached = self._up_to_stop = Structure(self.stop_idx, self.mask[:, :width] & before,
                                                  self.mask_G[:, :width] & before, default_device=self.default_device)
        return cached

    def clone(self) -> "Structure":
        return Structure(self.stop_idx.copy(), self.mask.copy(), self.mask_G.copy(), default_device=self.default_device)

    def __getitem__(self, index) -> "Structure":
        index = slice(index, index + 1) if isinstance(index, int) else index
        width = int(self.mask[index].sum(axis=1).max())
        return Structure(self.stop_idx[index], self.mask[index, :width], self.mask_G[index, :width],
                         default_device=self.default_device)

    @property
    def last_g_idx(self) -> np.ndarray:
        cols = np.broadcast_to(np.arange(self.mask.shape[1], dtype=self.stop_idx.dtype), self.mask.shape)
        return np.where(self.mask_G, cols, 0).argmax(axis=1)

    @property
    def mask_except_last(self) -> np.ndarray:
        out = self.mask.copy()
        out[np.arange(len(self)), self.last_g_idx + 1] = False
        return out


@dataclass
class Specs:
    structure: Structure
    epd: torch.Tensor
    hfov: torch.Tensor
    vig_up: Optional[torch.Tensor] = None
    vig_down: Optional[torch.Tensor] = None
    vig_x: Optional[torch.Tensor] = None

    def __post_init__(self):
        assert self.epd.dim() == 1, 'EPD should be 1-dimensional'
        assert self.hfov.dim() == 1, 'HFOV should be 1-dimensional'
        if self.vig_up is None or self.vig_down is None:
            self.vig_up = torch.zeros_like(self.epd)
            self.vig_down = torch.zeros_like(self.epd)
            self.vig_x = torch.zeros_like(self.epd)

    def __len__(self):
        return len(self.structure)

    def scale(self, factor) -> "Specs":
        return Specs(self.structure, self.epd * factor, self.hfov, self.vig_up, self.vig_down, self.vig_x)

    def up_to_stop(self) -> "Specs":
        return Specs(self.structure.up_to_stop(), self.epd, self.hfov, self.vig_up, self.vig_down, self.vig_x)

    def to(self, dtype) -> "Specs":
        return Specs(self.structure, self.epd.to(dtype), self.hfov.to(dtype), self.vig_up.to(dtype),
                     self.vig_down.to(dtype), self.vig_x.to(dtype))

    def double(self) -> "Specs":
        return self.to(torch.float64)

    def __getitem__(self, index) -> "Specs":
        index = slice(index, index + 1) if isinstance(index, int) else index
        return Specs(self.structure[index], self.epd[index], self.hfov[index], self.vig_up[index],
                     self.vig_down[index], self.vig_x[index])


# Results of the last few gradient-free conversions (padding of constant nd / v, dispersion of constant
# glasses): an optimisation loop re-builds the same Lens every step, and each of these is several tiny
# launches whose host cost (~8 us each) is what bounds small workloads.  Keyed by storage, version counter,
# shape and dtype, so an in-place update of the source invalidates the entry (writes through `.data` do not
# bump the version counter: they are the one way to defeat it, as they are for autograd itself).
_memo = {}
_MEMO_CAP = 8192


def _memoised(tag, extra, key_tensors, grad_tensors, fn):
    """fn() cached on (tag, extra, identity of key_tensors); recomputed whenever one of grad_tensors takes part
    in autograd (the result then carries a graph and must be fresh)."""
    if any(t.requires_grad for t in grad_tensors):
        return fn()
    # identity of a tensor's CONTENTS: storage address, layout (two views with the same first element and shape but
    # different strides -- M[0, :] and M[:, 0] -- are different data), version counter, dtype, device
    key = (tag, extra) + tuple((t.data_ptr(), t.storage_offset(), tuple(t.stride()), t._version, tuple(t.shape), t.dtype,
                                t.device) for t in key_tensors)
    hit = _memo.get(key)
    if hit is None or hit[0]._version != hit[1]:         # never computed, or somebody wrote into the cached result
        if len(_memo) > _MEMO_CAP:
            # drop the older half (dicts keep insertion order).  The cap is sized for a caller that loops over a minibatch of
            # lenses one at a time (optical_loss.py:96-110): every lens brings ~10 entries of a few hundred bytes, and a cap
            # below lenses x 10 means every call recomputes everything (64 did that to a 32-lens loop)
            for k in list(_memo)[: len(_memo) // 2]:
                del _memo[k]
        out = fn()
        # the entry keeps its key tensors alive: while it exists their storage cannot be freed and handed to
        # another tensor, so an equal (address, version, shape) key always means the same data
        hit = _memo[key] = (out, out._version, tuple(key_tensors))
    return hit[0]


def _pad_from_flat(flat: torch.Tensor, mask_t: torch.Tensor, fill: float, idx: torch.Tensor, what: str = "") -> torch.Tensor:
    """flat [n] (or [n, k]) -> padded [lens, row] (or [lens, row, k]) with `fill` elsewhere.  `what` names the
    field (c, t, nd ...): part of the memo key, so two fields padded alike never share an entry."""
    tail = tuple(flat.shape[1:])
    flat = flat.to(mask_t.device)
    if idx.numel() == mask_t.numel():            # nothing to pad (one lens, or equal-length lenses): a view
        return flat.reshape(*mask_t.shape, *tail)

    def pad():
        base = torch.full((mask_t.numel(), *tail), fill, dtype=flat.dtype, device=mask_t.device)
        return base.index_copy(0, idx, flat).reshape(*mask_t.shape, *tail)
    return _memoised("pad", (what, fill if fill == fill else "nan", tuple(mask_t.shape)), (flat, idx), (flat,), pad)


def _take(padded: torch.Tensor, width: int, idx: torch.Tensor) -> torch.Tensor:
    """The entries `idx` (flat positions in the [lens, width] layout) of padded[:, :width(, k)]."""
    cut = padded[:, :width]
    flat = cut.reshape(cut.shape[0] * width, *cut.shape[2:])
    if idx.numel() == flat.shape[0]:             # every entry is kept: a view
        return flat
    return _memoised("take", (width,), (padded, idx), (padded,), lambda: flat.index_select(0, idx))


@dataclass
class Lens:
    structure: Structure
    c: torch.Tensor
    t: torch.Tensor
    nd: torch.Tensor
    v: torch.Tensor
    # aspheric extension (not in the reference): conic constant per row and even polynomial terms
    # a4, a6, a8, a10 per row; 1-D / 2-D flat forms over the real rows, or padded [lens,row] / [lens,row,4]
    kappa: Optional[torch.Tensor] = None
    poly: Optional[torch.Tensor] = None

    def __post_init__(self):
        st = self.structure
        if (self.kappa is None) != (self.poly is None):
            n_rows = int(st.mask.sum())
            ref = self.kappa if self.kappa is not None else self.poly
            if self.kappa is None:
                self.kappa = torch.zeros(n_rows, dtype=ref.dtype, device=ref.device)
            else:
                self.poly = torch.zeros(n_rows, 4, dtype=ref.dtype, device=ref.device)
        if self.kappa is not None:
            if self.kappa.dim() == 1:
                self.kappa = _pad_from_flat(self.kappa, st.mask_torch, 0.0, st.idx_rows, "kappa")
            if self.poly.dim() == 2:                 # flat [rows, 4] -> padded [lens, row, 4]
                self.poly = _pad_from_flat(self.poly, st.mask_torch, 0.0, st.idx_rows, "poly")
        if self.c.dim() == 1:
            self.c = _pad_from_flat(self.c, st.mask_torch, 0.0, st.idx_rows, "c")
        if self.t.dim() == 1:
            self.t = _pad_from_flat(self.t, st.mask_torch, 0.0, st.idx_rows, "t")
        if self.nd.dim() == 1:
            self.nd = _pad_from_flat(self.nd, st.mask_G_torch, 1.0, st.idx_glass, "nd")
        if self.v.dim() == 1:
            self.v = _pad_from_flat(self.v, st.mask_G_torch, float('nan'), st.idx_glass, "v")

    def __len__(self):
        return len(self.structure)

    def _asph(self, fn):
        return (None, None) if self.kappa is None else (fn(self.kappa), fn(self.poly))

    def scale(self, factor) -> "Lens":
        if self.kappa is not None:
            powers = torch.tensor([3., 5., 7., 9.], dtype=self.poly.dtype, device=self.poly.device)
            return Lens(self.structure, self.c / factor, self.t * factor, self.nd, self.v, self.kappa,
                        self.poly / factor ** powers)
        return Lens(self.structure, self.c / factor, self.t * factor, self.nd, self.v)

    def up_to_stop(self) -> "Lens":
        st = self.structure.up_to_stop()
        n = st.mask.shape[1]
        kap, pol = self._asph(lambda a: _take(a, n, st.idx_rows))
        return Lens(st, _take(self.c, n, st.idx_rows), _take(self.t, n, st.idx_rows),
                    _take(self.nd, n, st.idx_glass), _take(self.v, n, st.idx_glass), kap, pol)

    def __getitem__(self, index) -> "Lens":
        index = slice(index, index + 1) if isinstance(index, int) else index
        st = self.structure[index]
        n = st.mask.shape[1]
        kap, pol = self._asph(lambda a: a[index, :n])
        return Lens(st, self.c[index, :n], self.t[index, :n], self.nd[index, :n], self.v[index, :n], kap, pol)

    def detach(self) -> "Lens":
        kap, pol = self._asph(lambda a: a.detach())
        return Lens(self.structure, self.c.detach(), self.t.detach(), self.nd.detach(), self.v.detach(), kap, pol)

    def to(self, dtype) -> "Lens":
        kap, pol = self._asph(lambda a: a.to(dtype))
        return Lens(self.structure, self.c.to(dtype), self.t.to(dtype), self.nd.to(dtype), self.v.to(dtype), kap, pol)

    def double(self) -> "Lens":
        return self.to(torch.float64)

    # ---- flat (compact) views -------------------------------------------------
    @property
    def flat_c(self):
        return self.c[self.structure.mask_torch]

    @flat_c.setter
    def flat_c(self, c):
        self.c = mask_replace(self.structure.mask, self.c, c)

    @property
    def flat_c_but_last(self):
        keep = self.structure.mask.copy()
        keep[np.arange(len(self)), self.structure.mask.sum(axis=1) - 1] = False
        return self.c[torch.from_numpy(keep).to(self.c.device)]

    @property
    def flat_t(self):
        return self.t[self.structure.mask_torch]

    @flat_t.setter
    def flat_t(self, t):
        self.t = mask_replace(self.structure.mask, self.t, t)

    @property
    def flat_nd(self):
        return self.nd[self.structure.mask_G_torch]

    @flat_nd.setter
    def flat_nd(self, nd):
        self.nd = mask_replace(self.structure.mask_G, self.nd, nd)

    @property
    def flat_v(self):
        return self.v[self.structure.mask_G_torch]

    @flat_v.setter
    def flat_v(self, v):
        self.v = mask_replace(self.structure.mask_G, self.v, v)

    # ---- dispersion -----------------------------------------------------------
    def get_refractive_indices(self, wavelengths) -> torch.Tensor:
        """n(lambda) = A + B / lambda^2 per row and wavelength [nm] -> [lens, row, W].

        A, B follow from nd and the Abbe number (lens_modeling.py:355-374).  Air rows give 1;
        rows whose Abbe number is 0 are dispersion-free (n = nd).
        """
        def dispersion():
            b = (self.nd - 1) / (self.v * (_LAMBDA_F ** -2 - _LAMBDA_C ** -2))
            a = self.nd - b / _LAMBDA_D ** 2
            lam = const_tensor(list(wavelengths), a.dtype, a.device, (1, 1, len(wavelengths)))
            n = a[..., None] + b[..., None] / lam ** 2
            n = torch.where(self.structure.mask_G_torch[..., None], n, torch.ones_like(n))
            dispersive = (self.v != 0)[..., None]
            return torch.where(dispersive, n, self.nd[..., None].expand_as(n))
        return _memoised("n", tuple(float(w) for w in wavelengths), (self.nd, self.v, self.structure.mask_G_torch),
                         (self.nd, self.v), dispersion)

    # ---- first-order properties (paraxial.py) ----------------------------------
    @property
    def efl(self):
        from . import paraxial
        return paraxial.get_first_order(self)[0]

    @property
    def bfl(self):
        from . import paraxial
        return paraxial.get_first_order(self)[1]

    @property
    def entrance_pupil_position(self):
        from . import paraxial
        return paraxial.compute_pupil_position(self)
