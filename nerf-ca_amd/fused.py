"""PyTorch-side plumbing around the C ABI: device buffers, streams and autograd seams.

``FieldBinding`` ties one coordinate network (an ``nn.Module`` owning ``nn.Parameter``s with the
reference's state-dict keys) to the library: it keeps the parameters in ONE flat f32 buffer in
``parameters()`` order (each ``nn.Parameter`` is a view into it), describes the net as an
``NcaNet`` and caches the MFMA-ordered weight image, re-packing only when a parameter changed.

``render_rays`` / ``eval_points`` are the two autograd entry points.  Nothing here computes the
path with torch ops: if the HIP library is missing these raise.
"""
from __future__ import annotations

import ctypes as C
import threading
import warnings
from typing import List, Optional, Sequence

import torch

from . import _capi
from ._capi import NcaNet, NcaRays, check, ptr


class PlanScope:
    """Planner options and the planner's record for ONE caller (a trainer, a test): every ray batch built inside ``with scope:`` carries
    ``scope.opts`` (NcaRays.plan_opts: they replace the process-wide tunables for those calls only) and has the library write what it
    decided into ``scope.plan`` (NcaRays.plan_out) -- so two trainers or threads of one process neither change nor read each other's
    plan.  The batch keeps its scope, so a backward on the autograd engine's thread uses its forward's options.

        with PlanScope(stage_fp8=0) as sc: ...; sc.decided()["bwd_kernel_mode"]
    """
    _tls = threading.local()

    def __init__(self, **opts):
        self.opts = _capi.NcaPlanOpts(**opts)
        self.plan = _capi.NcaPlan()

    def __enter__(self):
        stack = getattr(PlanScope._tls, "stack", None)
        if stack is None:
            stack = PlanScope._tls.stack = []
        stack.append(self)
        return self

    def __exit__(self, *exc):
        PlanScope._tls.stack.pop()
        return False

    @staticmethod
    def current() -> Optional["PlanScope"]:
        stack = getattr(PlanScope._tls, "stack", None)
        return stack[-1] if stack else None

    def decided(self) -> dict:
        return _capi.plan_dict(self.plan)

_ACT = {"softplus": _capi.ACT_SOFTPLUS, "clamp": _capi.ACT_CLAMP}  # anything else -> sigmoid (model_helpers.py:63-70)

# upper bound on the backward workspace (the output gradients -- and, without a forward store, the layer inputs -- of one
# ray chunk).  Fewer, larger chunks are faster (one chunk at the bench size: -3.7 % step time against 6 GiB chunks) and an
# MI355X has 288 GB; if the allocation fails the bound is halved until it fits.
BWD_WORKSPACE_BYTES = 40 << 30

_PREC = {"f32": _capi.PREC_F32, "fp32": _capi.PREC_F32, "bf16": _capi.PREC_BF16}


def set_precision(prec: str, *models) -> None:
    """Select the arithmetic of the MLP contractions for the given drop-in models: ``"f32"`` (parity
    mode: f32 MFMA, 1e-5 vs the reference) or ``"bf16"`` (throughput mode: bf16 MFMA operands, f32
    accumulation and f32 master weights; PSNR-gated, not 1e-5-gated)."""
    code = _PREC[prec]
    for m in models:
        b = m._binding
        if b.prec != code:
            b.prec = code
            b.packed = None


# Debug switch (tests/test_soak_bench_path.py, NERFCA_POISON=1): every scratch buffer handed to the library -- forward workspace and
# store, backward workspace, loss workspace -- is filled with a NaN pattern first (f32 NaN, bf16 NaN pairs, large e4m3 / e5m2
# bytes), eagerly or as part of a captured graph.  The library never reads a byte it has not written, so results must not change.
import os as _os
POISON_BUFFERS = _os.environ.get("NERFCA_POISON") == "1"


def _scratch(nbytes: int, dev) -> torch.Tensor:
    t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    if POISON_BUFFERS and nbytes >= 4:
        t[: nbytes // 4 * 4].view(torch.int32).fill_(0x7FC12345)
    return t


def _alloc_workspace(size_for_cap, dev):
    """Allocate the backward workspace the library sizes for a byte cap (the configured bound, at most 80 % of what the device can
    still give); halve the cap while the allocation fails."""
    cap = max(1 << 20, min(BWD_WORKSPACE_BYTES, int(0.8 * _usable_bytes(dev))))
    while True:
        wbytes = size_for_cap(cap)
        try:
            return _scratch(wbytes, dev), wbytes
        except torch.cuda.OutOfMemoryError:
            if cap <= (1 << 30):
                raise
            cap >>= 1


def act_code(name: str) -> int:
    return _ACT.get(name, _capi.ACT_SIGMOID)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _require_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise _capi.NcaError(f"{what} must live on the GPU: the fused NeRF-CA path has no CPU implementation "
                             f"(got device {t.device})")


class FieldBinding:
    """Flat parameters + packed weight image of one network."""

    def __init__(self, module: torch.nn.Module, net: NcaNet, prec: int = _capi.PREC_F32):
        self.module = module
        self.net = net
        self.prec = prec
        self.flat: Optional[torch.Tensor] = None
        self.packed: Optional[torch.Tensor] = None
        self.static_window: Optional[torch.Tensor] = None   # device f32[L] that overrides the module's band window
        self._offsets: List[int] = []
        self._pads: List[tuple] = []                        # (padded shape, logical shape) per parameter
        self._strides: List[tuple] = []                     # the parameter views' strides
        self.gaps: List[tuple] = []                         # (offset, count) of bias slots of a module built with use_bias=False
        self.reflatten()

    # -- parameters ---------------------------------------------------------------------------
    def params(self) -> List[torch.nn.Parameter]:
        return list(self.module.parameters())

    def n_params(self) -> int:
        return sum(p.numel() for p in self.module.parameters())

    def reflatten(self) -> None:
        """(Re)create the flat buffer on the parameters' current device and re-point every
        ``nn.Parameter`` at its slice.  Called at construction and after ``module.to(...)``.

        The buffer is in the library's natural order (``[time_latents,] weight, bias, weight, bias, ...``).  A module built with
        ``use_bias=False`` (CPPN.py:15-19) owns no bias parameters: their slots stay in the buffer as zeros (``self.gaps``) -- a
        layer without a bias is a layer whose bias is zero and never updated -- and the kernels' gradients for them are dropped."""
        ps = self.params()
        if not ps:
            return
        dev = ps[0].device
        biasless = not getattr(self.module, "use_bias", True)
        F = int(getattr(self.module, "num_filters", self.net.F))
        Fp = int(self.net.F)                          # the kernels' width: F rounded up to 32 / 64 / 128, or to a multiple of 16 on the general kernels (model/_field.py)
        # (parameter, logical shape, padded shape) per slot of the library's natural order.  A net narrower than the kernels' width
        # runs as the wider net whose extra units have zero weights and biases: they stay at relu(0) = 0, feed nothing and receive
        # zero gradients, so no optimiser ever moves them; every parameter is the leading [rows, columns] block of its padded matrix.
        slots = []
        for name, p in self.module.named_parameters():
            if name.endswith(".weight"):
                out, inn = p.shape
                first, last = name.startswith("early_pts_layers.0."), name.startswith("output_linear.")
                slots.append((p, (out, inn), (out if last else Fp, inn if first else inn + (Fp - F))))
                if biasless:
                    slots.append((None, None, (out if last else Fp,)))
            elif name.endswith(".bias"):
                slots.append((p, tuple(p.shape), (p.shape[0] if name.startswith("output_linear.") else Fp,)))
            else:
                slots.append((p, tuple(p.shape), tuple(p.shape)))
        numel = lambda shp: int(torch.Size(shp).numel())
        flat = torch.zeros(sum(numel(pad) for _, _, pad in slots), dtype=torch.float32, device=dev)
        self._offsets, self._pads, self._strides, self.gaps = [], [], [], []
        off = 0
        with torch.no_grad():
            for p, shp, pad in slots:
                n = numel(pad)
                if p is None:
                    self.gaps.append((off, n))
                else:
                    view = flat[off:off + n].view(pad)[tuple(slice(0, d) for d in shp)]
                    view.copy_(p.detach().to(torch.float32))
                    p.data = view
                    self._offsets.append(off)
                    self._pads.append((pad, shp))
                    self._strides.append(tuple(view.stride()))
                off += n
        self.flat = flat
        self.packed = None

    def zero_gaps(self, t: torch.Tensor) -> None:
        """Zero the slots of ``t`` (a flat tensor in the buffer's layout) that belong to no parameter."""
        for off, n in self.gaps:
            t[off:off + n].zero_()

    def _is_flat(self) -> bool:
        if self.flat is None:
            return False
        base = self.flat.data_ptr()
        for p, off, st in zip(self.params(), self._offsets, self._strides):
            if p.data_ptr() != base + 4 * off or p.dtype != torch.float32 or p.stride() != st:
                return False
        return True

    def ensure_packed(self) -> torch.Tensor:
        """Re-order the current parameters into the kernels' weight images.  Done on EVERY forward:
        optimisers update parameters in place without a reliable change signal (fused Adam does not
        bump tensor version counters), and the pack kernel costs ~10 us."""
        if not self._is_flat():
            self.reflatten()
        _require_cuda(self.flat, "network parameters")
        lib = _capi.lib()
        if self.packed is None or self.packed.device != self.flat.device:
            expect = check(lib.nca_param_count(C.byref(self.net)))
            if expect != self.flat.numel():
                raise _capi.NcaError(f"parameter count mismatch: module has {self.flat.numel()}, descriptor expects {expect}")
            nbytes = check(lib.nca_packed_bytes(C.byref(self.net), self.prec))
            self.packed = torch.empty(nbytes, dtype=torch.uint8, device=self.flat.device)
        check(lib.nca_pack_weights(C.byref(self.net), ptr(self.flat), ptr(self.packed), self.prec, _stream()))
        return self.packed

    @staticmethod
    def ensure_packed_pair(bs: "FieldBinding", bd: "FieldBinding"):
        """``ensure_packed`` of both nets of a composite render in ONE launch (nca_pack_weights2)."""
        if bs.prec != bd.prec or bs is bd:
            return bs.ensure_packed(), bd.ensure_packed()
        lib = _capi.lib()
        for b in (bs, bd):
            if not b._is_flat():
                b.reflatten()
            _require_cuda(b.flat, "network parameters")
            if b.packed is None or b.packed.device != b.flat.device:
                expect = check(lib.nca_param_count(C.byref(b.net)))
                if expect != b.flat.numel():
                    raise _capi.NcaError(f"parameter count mismatch: module has {b.flat.numel()}, descriptor expects {expect}")
                b.packed = torch.empty(check(lib.nca_packed_bytes(C.byref(b.net), b.prec)), dtype=torch.uint8, device=b.flat.device)
        check(lib.nca_pack_weights2(C.byref(bs.net), ptr(bs.flat), ptr(bs.packed), C.byref(bd.net), ptr(bd.flat), ptr(bd.packed), bs.prec, _stream()))
        return bs.packed, bd.packed

    def split_grads(self, gflat: torch.Tensor) -> List[torch.Tensor]:
        out = []
        for off, (pad, shp) in zip(self._offsets, self._pads):
            n = int(torch.Size(pad).numel())
            out.append(gflat[off:off + n].view(pad)[tuple(slice(0, d) for d in shp)])
        return out


def _f32c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    return t.detach().to(torch.float32).contiguous()


class _RayBatch:
    """Device-side, dtype-normalised view of the arguments of obtain_train_predictions_iter."""

    def __init__(self, origins, directions, phases, I0, z, dists, act: str, single: bool, scale: float):
        _require_cuda(origins, "ray origins")
        self.R = origins.shape[0]
        self.f64 = origins.dtype == torch.float64
        rdt = torch.float64 if self.f64 else torch.float32
        self.o = origins.detach().to(rdt).contiguous()
        self.d = directions.detach().to(rdt).contiguous()
        dev = origins.device
        self.z = z.detach().to(device=dev, dtype=torch.float32).contiguous()
        self.S = self.z.shape[-1]
        self.z_stride_r = 0 if self.z.dim() == 1 else self.S
        self.dists = dists.detach().to(device=dev, dtype=torch.float64).contiguous()
        self.I0 = I0.detach().to(device=dev, dtype=torch.float32).contiguous()
        self.ph = None
        self.ps_r = self.ps_s = 0
        if phases is not None:
            ph = phases.detach().to(device=dev, dtype=torch.int32)
            if ph.dim() == 1:            # one id per ray
                ph = ph.contiguous()
                self.ps_r, self.ps_s = 1, 0
            else:                        # [R, S] as run_composite.py:265 builds it
                ph = ph.reshape(self.R, -1).contiguous()
                self.ps_r, self.ps_s = ph.shape[1], 1
            self.ph = ph
        self.act, self.single, self.scale = act_code(act), 1 if single else 0, float(scale)
        self.scope = PlanScope.current()          # the caller's planner options / record (None: the process-wide ones)

    def desc(self, store_format: int = 0) -> NcaRays:
        sc = self.scope
        return NcaRays(R=self.R, S=self.S, ray_is_f64=1 if self.f64 else 0, origins=ptr(self.o), dirs=ptr(self.d),
                       phase=ptr(self.ph), phase_stride_r=self.ps_r, phase_stride_s=self.ps_s, z=ptr(self.z),
                       z_stride_r=self.z_stride_r, dists=ptr(self.dists), I0=ptr(self.I0), act=self.act,
                       single_field=self.single, scale=self.scale, store_format=store_format,
                       plan_opts=C.addressof(sc.opts) if sc is not None else None, plan_out=C.addressof(sc.plan) if sc is not None else None)


# A forward that will be followed by a backward leaves every layer input, the ReLU masks and the raw outputs in a
# store (what the reference's autograd graph keeps); the backward then skips the recompute.  1.6 - 2.8 KB per sample in
# bf16, 5.9 KB in f32: the store is used when it fits under this limit AND under the device's free memory (below),
# otherwise the backward recomputes (set to 0 to always recompute).
STORE_FORWARD_LIMIT_BYTES = 96 << 30
# how often a backward had to fall back to the recompute path because the store did not fit or could not be allocated: a bench line
# taken with a non-zero count was not timed on the stored path.  Every fallback WARNS (once per size); under NERFCA_STRICT=1 (or
# STRICT_STORE = True) it is an error -- bench.py runs strict, so that it can never silently time the recompute path.
STORE_FALLBACKS = 0
STRICT_STORE: Optional[bool] = None          # None: follow the environment (NERFCA_STRICT=1)


def _strict_store() -> bool:
    import os
    return bool(STRICT_STORE) if STRICT_STORE is not None else os.environ.get("NERFCA_STRICT") == "1"


class StoreFallbackWarning(RuntimeWarning):
    pass


def _usable_bytes(dev) -> int:
    """Bytes this process can still take on `dev`: the driver's free memory plus what torch's caching allocator holds unused."""
    free, _ = torch.cuda.mem_get_info(dev)
    return int(free + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev))


def store_limit_bytes(dev) -> int:
    """The forward-store limit that applies now: the configured cap, and at most 45 % of what the device can still give (a
    backward workspace of similar size follows, and the fine pass holds two stores at once)."""
    if STORE_FORWARD_LIMIT_BYTES <= 0:
        return 0
    return min(STORE_FORWARD_LIMIT_BYTES, int(0.45 * _usable_bytes(dev)))


def _same_window(bs: FieldBinding, bd: FieldBinding) -> bool:
    """Both nets use the same band encoding with equal window VALUES this step (the composite.txt default)."""
    if bs.net.enc_mode != _capi.ENC_BANDS or bd.net.enc_mode != _capi.ENC_BANDS or bs.net.L != bd.net.L:
        return False
    if bs.static_window is not None or bd.static_window is not None:
        return False       # graph-owned vectors: the owner decides (pointer equality is fixed at capture)
    a, b = bs.module._band_window(), bd.module._band_window()
    return (not a.is_cuda) and (not b.is_cuda) and a.shape == b.shape and bool(torch.equal(a, b))


def forward_store_bytes(batch: _RayBatch, bs: FieldBinding, bd: Optional[FieldBinding]) -> int:
    """Bytes a storing forward of this ray batch would leave for its backward (0: no store for this configuration)."""
    desc = batch.desc()
    return check(_capi.lib().nca_render_store_bytes(C.byref(desc), C.byref(bs.net), C.byref(bd.net) if bd is not None else None, bs.prec))


class RaySums:
    """What a forward called with ``want_pix=False`` returns in place of pix: the per-tile ray sums it left in its workspace (f64[R][nchunk])
    and the rays' I0 -- ``fused_losses`` hands them to the loss kernel, which forms pix itself (one launch less per step)."""

    def __init__(self, part: torch.Tensor, nchunk: int, I0: torch.Tensor):
        self.part, self.nchunk, self.I0 = part, nchunk, I0


def render_forward_raw(batch: _RayBatch, bs: FieldBinding, bd: Optional[FieldBinding], for_backward: bool = False, want_pix: bool = True):
    """Fused forward without autograd: returns (pix f64[R], sigma_s, sigma_d | None, keep) where ``keep``
    pins the packed weights / encoding buffers (and, with ``for_backward``, the forward store) the matching
    backward must see.  ``want_pix=False`` (nets of one width): pix is a ``RaySums`` for ``fused_losses`` instead of a tensor."""
    lib = _capi.lib()
    dev = batch.o.device
    if bd is not None:
        packed_s, packed_d = FieldBinding.ensure_packed_pair(bs, bd)
    else:
        packed_s, packed_d = bs.ensure_packed(), None
    win_s, four_s = bs.module._enc_buffers()
    win_d, four_d = bd.module._enc_buffers() if bd is not None else (None, None)
    if bd is not None and win_s is not None and win_d is not None and win_s.data_ptr() != win_d.data_ptr() and _same_window(bs, bd):
        win_d = win_s      # one vector for both nets: the library then stores the encoded input once (see share_enc)
    R, S = batch.R, batch.S
    general = _capi.net_is_general(bs.net) or (bd is not None and _capi.net_is_general(bd.net))
    if general or (bd is not None and bd.net.F != bs.net.F):
        want_pix = True          # (nets of different width, or on the general kernels, composite in a kernel of their own)
    pix = torch.empty(R, dtype=torch.float64, device=dev) if want_pix else None
    sig_s = torch.empty((R, S), dtype=torch.float32, device=dev)
    sig_d = torch.empty((R, S), dtype=torch.float32, device=dev) if bd is not None else None
    desc = batch.desc()
    if general:     # a chunk of activations on top of the ray sums (include/nerfca_hip.h: nca_render_fwd_workspace_nets)
        net_d_ = C.byref(bd.net) if bd is not None else None
        work, wbytes = _alloc_workspace(lambda cap: check(lib.nca_render_fwd_workspace_nets(C.byref(desc), C.byref(bs.net), net_d_, bs.prec, cap)), dev)
    else:
        wbytes = check(lib.nca_render_fwd_workspace(C.byref(desc)))
        work = _scratch(wbytes, dev)
    store = None
    global STORE_FALLBACKS
    if for_backward and STORE_FORWARD_LIMIT_BYTES > 0:
        sbytes = check(lib.nca_render_store_bytes(C.byref(desc), C.byref(bs.net), C.byref(bd.net) if bd is not None else None, bs.prec))
        if 0 < sbytes <= store_limit_bytes(dev):
            try:
                store = _scratch(sbytes, dev)
            except torch.cuda.OutOfMemoryError:        # not enough free HBM for the store: the backward recomputes instead
                store = None
        if sbytes > 0 and store is None:
            STORE_FALLBACKS += 1
            msg = (f"the forward store of this batch ({sbytes / 2**30:.1f} GiB; limit {store_limit_bytes(dev) / 2**30:.1f} GiB) was not allocated: "
                   "this backward recomputes the layers (slower; same results in f32, the recompute arithmetic in bf16)")
            if _strict_store():
                raise _capi.NcaError(msg + " -- refused under NERFCA_STRICT=1 / fused.STRICT_STORE")
            warnings.warn(msg, StoreFallbackWarning, stacklevel=2)
    # (the return value names what the forward left in the store -- the planner's choice of staging for this batch; the backward
    # is told through NcaRays.store_format, so a change of the process-wide options in between cannot reinterpret the bytes)
    fmt = check(lib.nca_render_fwd(C.byref(desc), bs.prec,
                                   C.byref(bs.net), ptr(packed_s), ptr(win_s), ptr(four_s),
                                   C.byref(bd.net) if bd is not None else None, ptr(packed_d), ptr(win_d), ptr(four_d),
                                   ptr(bd.flat) if bd is not None else None,
                                   ptr(pix), ptr(sig_s), ptr(sig_d), ptr(work), wbytes,
                                   ptr(store), store.numel() if store is not None else 0, _stream()))
    if fmt == _capi.STORE_NONE:
        store = None          # the planner wrote no store (e.g. the options changed since it was sized): the backward recomputes
    if not want_pix:
        tile = 64 if bs.prec == _capi.PREC_BF16 else 32
        pix = RaySums(work, (S + tile - 1) // tile, batch.I0)
    return pix, sig_s, sig_d, (packed_s, packed_d, win_s, four_s, win_d, four_d, store, fmt)


def render_backward_raw(batch: _RayBatch, bs: FieldBinding, bd: Optional[FieldBinding], keep, g_pix, g_sig_s, g_sig_d, want_depth_grad: bool = False,
                        out_s: Optional[torch.Tensor] = None, out_d: Optional[torch.Tensor] = None):
    """Fused backward (recompute + dgrad + wgrad + reduce): returns flat f32 gradients per net (and, with ``want_depth_grad``,
    d loss / d depth f32[R,S] as a third value: the f32 path's nca_render_bwd_depth).  ``out_s`` / ``out_d``: caller-owned contiguous f32
    buffers the gradients are written into (slices of ONE flat buffer: the step's all-reduce and Adam then need no concatenation)."""
    lib = _capi.lib()
    packed_s, packed_d, win_s, four_s, win_d, four_d, store, fmt = keep
    dev = batch.o.device
    gp = torch.zeros(batch.R, dtype=torch.float64, device=dev) if g_pix is None else g_pix.detach().to(torch.float64).contiguous()
    gs, gd = _f32c(g_sig_s), _f32c(g_sig_d)
    for o_, b_ in ((out_s, bs), (out_d, bd)):
        if o_ is not None and (b_ is None or o_.numel() != b_.flat.numel() or o_.dtype != torch.float32 or not o_.is_contiguous() or o_.device != dev):
            raise _capi.NcaError("out_s / out_d must be contiguous f32 buffers of the nets' flat parameter counts on the rays' device")
    grads_s = out_s if out_s is not None else torch.empty(bs.flat.numel(), dtype=torch.float32, device=dev)
    grads_d = (out_d if out_d is not None else torch.empty(bd.flat.numel(), dtype=torch.float32, device=dev)) if bd is not None else None
    desc = batch.desc(store_format=fmt if store is not None else 0)
    net_d = C.byref(bd.net) if bd is not None else None
    work, wbytes = _alloc_workspace(lambda cap: check(lib.nca_render_bwd_workspace(C.byref(desc), C.byref(bs.net), net_d, bs.prec, cap)), dev)
    if want_depth_grad:
        g_depth = torch.zeros((batch.R, batch.S), dtype=torch.float32, device=dev)
        check(lib.nca_render_bwd_depth(C.byref(desc), bs.prec,
                                       C.byref(bs.net), ptr(packed_s), ptr(win_s), ptr(four_s), ptr(bs.flat),
                                       net_d, ptr(packed_d), ptr(win_d), ptr(four_d), ptr(bd.flat) if bd is not None else None,
                                       ptr(gp), ptr(gs), ptr(gd), ptr(grads_s), ptr(grads_d), ptr(g_depth), ptr(work), wbytes,
                                       ptr(store), store.numel() if store is not None else 0, _stream()))
        return grads_s, grads_d, g_depth
    check(lib.nca_render_bwd(C.byref(desc), bs.prec,
                             C.byref(bs.net), ptr(packed_s), ptr(win_s), ptr(four_s), ptr(bs.flat),
                             net_d, ptr(packed_d), ptr(win_d), ptr(four_d), ptr(bd.flat) if bd is not None else None,
                             ptr(gp), ptr(gs), ptr(gd), ptr(grads_s), ptr(grads_d), ptr(work), wbytes,
                             ptr(store), store.numel() if store is not None else 0, _stream()))
    return grads_s, grads_d


class _RenderFn(torch.autograd.Function):
    """pix, sigma_s, sigma_d = fused(rays; depths, dists; params_s, params_d).

    ``z_in`` / ``dists_in`` are the tensors the caller passed (their values already sit in ``batch``): they are inputs only so
    that autograd can ask for their gradients -- the reference's fine pass differentiates through its sampled depths and
    through the ray-0 interval lengths (model_helpers.py:146-158)."""

    @staticmethod
    def forward(ctx, batch: _RayBatch, bs: FieldBinding, bd: Optional[FieldBinding], n_s: int, z_in, dists_in, *params):
        # batch.record: autograd is recording this call (see render_rays) -- only then will a backward follow and only
        # then is the forward asked to keep its layer inputs
        pix, sig_s, sig_d, keep = render_forward_raw(batch, bs, bd, for_backward=getattr(batch, "record", False))
        ctx.batch, ctx.bs, ctx.bd, ctx.keep = batch, bs, bd, keep
        ctx.z_shape = None if z_in is None else tuple(z_in.shape)
        ctx.z_dtype = None if z_in is None else z_in.dtype
        ctx.dists_dtype = None if dists_in is None else dists_in.dtype
        if dists_in is not None and dists_in.requires_grad:
            ctx.sig = (sig_s, sig_d)
        if not batch.f64:
            pix = pix.to(torch.float32)
        if bd is None:
            return pix, sig_s
        return pix, sig_s, sig_d

    @staticmethod
    def backward(ctx, g_pix, g_sig_s, g_sig_d=None):
        bs, bd, batch = ctx.bs, ctx.bd, ctx.batch
        want_z, want_dists = ctx.needs_input_grad[4], ctx.needs_input_grad[5]
        res = render_backward_raw(batch, bs, bd, ctx.keep, g_pix, g_sig_s, g_sig_d, want_depth_grad=want_z)
        grads_s, grads_d = res[0], res[1]
        g_z = g_dists = None
        if want_z:
            g_z = res[2]
            if len(ctx.z_shape) == 1:                 # one depth vector shared by all rays
                g_z = g_z.sum(0)
            g_z = g_z.to(ctx.z_dtype)
        if want_dists:
            # pix = I0 - sum_s (sigma_s + sigma_d) dists  (scaled sigmas; single field: sigma * scale), model_helpers.py:80-82, 92-95
            sig_s, sig_d = ctx.sig
            gp = torch.zeros(batch.R, dtype=torch.float64, device=sig_s.device) if g_pix is None else g_pix.to(torch.float64)
            tot = (sig_s.double() * batch.scale) if bd is None else (sig_s + sig_d).double()
            g_dists = -(gp[:, None] * tot).sum(0).to(ctx.dists_dtype)
        bs.last_grad = grads_s
        out = bs.split_grads(grads_s)
        if bd is not None:
            bd.last_grad = grads_d
            out = out + bd.split_grads(grads_d)
        return (None, None, None, None, g_z, g_dists, *out)


def fused_losses(pix, gt, wpix, sig_s, sig_d, dists, run_args, weights, inv_R=None, want_grads=True, weights_dev=None, unit_mse=False,
                 want_dists_grad=False, terms_f32: Optional[torch.Tensor] = None, pix_out: Optional[torch.Tensor] = None):
    """weighted MSE + compute_losses + the loss assembly of run_composite.py:287-292 in one HIP pass.

    ``weights`` = (favor_s_weight, dynamic_entro_weight, occl_weight, l1_weight) of this step.
    Returns ``(terms f64[13] on device, g_pix f64[R], g_sigma_s f32[R,S], g_sigma_d f32[R,S])``; see
    ``_capi.TERM_NAMES`` for the order of ``terms``.  ``inv_R`` = 1 / global ray count (default 1/R).
    ``weights_dev`` (device f64[4]) replaces ``weights`` with values the kernels read at run time, which is
    what a captured HIP graph needs.  ``unit_mse``: the pixel term uses unit weights while the regularisers keep
    ``wpix`` (the fine pass's ``weighted_pixs_ones``, run_composite.py:296-299).  ``want_dists_grad``: a fifth return value,
    d loss / d dists f64[S] (the fine pass differentiates through ray 0's interval lengths, model_helpers.py:150).
    """
    lib = _capi.lib()
    _require_cuda(sig_s, "sigma")
    dev = sig_s.device
    R, S = sig_s.shape
    f64 = lambda t: t.detach().to(device=dev, dtype=torch.float64).contiguous()
    sums = pix if isinstance(pix, RaySums) else None
    pix = None if sums is not None else f64(pix)
    gt, wpix, dists = f64(gt), f64(wpix), f64(dists)
    ss, sd = _f32c(sig_s), _f32c(sig_d)
    desc = _capi.NcaLoss(R=R, S=S, use_weighting=1 if run_args.entro_use_weighting else 0, skew=float(run_args.skewness_val),
                         mask_thre=float(run_args.entro_mask_thre), weighted_thresh=float(run_args.entro_weighted_thresh),
                         w_favor=float(weights[0]), w_dent=float(weights[1]), w_occl=float(weights[2]), w_l1=float(weights[3]),
                         inv_R=float(inv_R if inv_R is not None else 1.0 / R), weights_dev=None, unit_mse=1 if unit_mse else 0, reserved=0,
                         g_dists=None, dists_work=None, term_grads=None)
    if weights_dev is not None:
        if weights_dev.dtype != torch.float64 or weights_dev.numel() != 4 or not weights_dev.is_cuda or not weights_dev.is_contiguous():
            raise _capi.NcaError("weights_dev must be a contiguous device f64[4]")
        desc.weights_dev = ptr(weights_dev)
    if sums is not None:
        if sums.part.numel() * sums.part.element_size() < R * sums.nchunk * 8 or sums.I0.shape[0] != R:
            raise _capi.NcaError("RaySums does not belong to this batch")
        desc.ray_part, desc.ray_I0, desc.ray_nchunk = ptr(sums.part), ptr(sums.I0), int(sums.nchunk)
        if pix_out is not None:
            if pix_out.dtype != torch.float64 or pix_out.numel() != R or not pix_out.is_contiguous() or pix_out.device != dev:
                raise _capi.NcaError("pix_out must be a contiguous device f64[R]")
            desc.pix_out = ptr(pix_out)
    if terms_f32 is not None:
        if terms_f32.dtype != torch.float32 or terms_f32.numel() != len(_capi.TERM_NAMES) or not terms_f32.is_contiguous() or terms_f32.device != dev:
            raise _capi.NcaError("terms_f32 must be a contiguous device f32[13]")
        desc.terms_f32 = ptr(terms_f32)
    terms = torch.empty(len(_capi.TERM_NAMES), dtype=torch.float64, device=dev)
    g_pix = g_s = g_d = None
    if want_grads:
        g_pix = torch.empty(R, dtype=torch.float64, device=dev)
        g_s = torch.empty((R, S), dtype=torch.float32, device=dev)
        g_d = torch.empty((R, S), dtype=torch.float32, device=dev)
    wbytes = check(lib.nca_loss_workspace(R))
    work = _scratch(wbytes, dev)
    g_dists = dwork = None
    if want_dists_grad:
        if not want_grads:
            raise _capi.NcaError("want_dists_grad needs want_grads")
        g_dists = torch.empty(S, dtype=torch.float64, device=dev)
        dwork = _scratch(R * S * 8, dev)
        desc.g_dists, desc.dists_work = ptr(g_dists), ptr(dwork)
    check(lib.nca_loss_fwd_bwd(C.byref(desc), ptr(pix), ptr(gt), ptr(wpix), ptr(ss), ptr(sd), ptr(dists), ptr(terms),
                               ptr(g_pix), ptr(g_s), ptr(g_d), ptr(work), wbytes, _stream()))
    if want_dists_grad:
        return terms, g_pix, g_s, g_d, g_dists
    return terms, g_pix, g_s, g_d


class _LossTermsFn(torch.autograd.Function):
    """compute_losses (train/model_helpers.py:250-262) as ONE autograd node over the HIP loss kernel: forward = the kernel's values,
    backward = the kernel in term-gradient mode (NcaLoss.term_grads: the eleven upstream scalars weight the terms' gradients), so a
    script that keeps the reference's own loss assembly (train/run_composite.py:287-292) launches two kernels per direction where
    the reference's torch functions launch ~25 elementwise / reduction kernels over [R, S]."""

    @staticmethod
    def _launch(sig_s, sig_d, dists64, wpix64, opts, term_grads, want_grads, want_dists_grad):
        lib = _capi.lib()
        dev = sig_s.device
        R, S = sig_s.shape
        use_w, skew, mask_thre, w_thresh = opts
        desc = _capi.NcaLoss(R=R, S=S, use_weighting=1 if use_w else 0, skew=float(skew), mask_thre=float(mask_thre), weighted_thresh=float(w_thresh),
                             w_favor=0.0, w_dent=0.0, w_occl=0.0, w_l1=0.0, inv_R=1.0 / R, weights_dev=None, unit_mse=0, reserved=0,
                             g_dists=None, dists_work=None, term_grads=ptr(term_grads))
        terms = torch.empty(len(_capi.TERM_NAMES), dtype=torch.float64, device=dev)
        g_s = g_d = g_dists = dwork = None
        if want_grads:
            g_s = torch.empty((R, S), dtype=torch.float32, device=dev)
            g_d = torch.empty((R, S), dtype=torch.float32, device=dev)
            if want_dists_grad:
                g_dists = torch.empty(S, dtype=torch.float64, device=dev)
                dwork = _scratch(R * S * 8, dev)
                desc.g_dists, desc.dists_work = ptr(g_dists), ptr(dwork)
        wbytes = check(lib.nca_loss_workspace(R))
        work = _scratch(wbytes, dev)
        check(lib.nca_loss_fwd_bwd(C.byref(desc), None, None, ptr(wpix64), ptr(sig_s), ptr(sig_d), ptr(dists64), ptr(terms),
                                   None, ptr(g_s), ptr(g_d), ptr(work), wbytes, _stream()))
        return terms, g_s, g_d, g_dists

    @staticmethod
    def forward(ctx, sig_s, sig_d, dists, wpix, opts):
        dev = sig_s.device
        ss, sd = _f32c(sig_s), _f32c(sig_d)
        d64 = dists.detach().to(device=dev, dtype=torch.float64).contiguous()
        w64 = wpix.detach().to(device=dev, dtype=torch.float64).contiguous()
        zeros = torch.zeros(11, dtype=torch.float64, device=dev)
        terms, _, _, _ = _LossTermsFn._launch(ss, sd, d64, w64, opts, zeros, False, False)
        ctx.save_for_backward(ss, sd, d64, w64)
        ctx.opts, ctx.dists_dtype, ctx.sig_dtypes = opts, dists.dtype, (sig_s.dtype, sig_d.dtype)
        T = _capi.TERM_NAMES
        lo = sig_s.dtype                                            # blend-weight terms and maxima stay in sigma's dtype (f32)
        hi = torch.promote_types(sig_s.dtype, dists.dtype)          # everything that touches dists takes the promoted dtype (f64 in the real script)
        pick = lambda name, dt: terms[T.index(name)].to(dt)
        out = (pick("blendw", lo), pick("sigma_s_max", lo), pick("sigma_d_max", lo), pick("favor_s", lo), pick("s_entropy", hi), pick("s_entropy_sum", hi),
               pick("d_entropy", hi), pick("d_entropy_sum", hi), pick("d_occl", hi), pick("s_l1", hi), pick("s_l2", hi))
        ctx.mark_non_differentiable(out[1], out[2])
        return out

    @staticmethod
    def backward(ctx, *g):
        ss, sd, d64, w64 = ctx.saved_tensors
        dev = ss.device
        tg = torch.stack([torch.zeros((), dtype=torch.float64, device=dev) if gi is None else gi.detach().to(torch.float64).reshape(()) for gi in g])
        want_d = ctx.needs_input_grad[2]
        _, g_s, g_d, g_dists = _LossTermsFn._launch(ss, sd, d64, w64, ctx.opts, tg, True, want_d)
        return (g_s.to(ctx.sig_dtypes[0]) if ctx.needs_input_grad[0] else None, g_d.to(ctx.sig_dtypes[1]) if ctx.needs_input_grad[1] else None,
                g_dists.to(ctx.dists_dtype) if want_d else None, None, None)


def loss_terms(static_sigma, temp_sigma, dists, weighted_pixs, run_args):
    """The reference's 11-tuple of compute_losses (train/model_helpers.py:250-262) from the HIP loss kernel, differentiable w.r.t.
    both density fields and the interval lengths.  GPU tensors only: there is no torch implementation behind it."""
    _require_cuda(static_sigma, "static_sigma")
    _require_cuda(temp_sigma, "temp_sigma")
    if static_sigma.dim() != 2 or static_sigma.shape != temp_sigma.shape or dists.dim() != 1 or dists.shape[0] != static_sigma.shape[1]:
        raise _capi.NcaError(f"compute_losses: expected sigma [R,S] twice and dists [S], got {tuple(static_sigma.shape)}, {tuple(temp_sigma.shape)}, {tuple(dists.shape)}")
    R = static_sigma.shape[0]
    use_w = bool(run_args.entro_use_weighting) and weighted_pixs is not None and len(weighted_pixs) > 0
    if use_w and weighted_pixs.shape[0] > R:
        raise _capi.NcaError(f"compute_losses: {weighted_pixs.shape[0]} pixel weights for {R} rays")
    wp = weighted_pixs if use_w else torch.ones(R, dtype=torch.float64, device=static_sigma.device)
    if use_w and wp.shape[0] < R:
        # fewer weights than rays: the reference fills weighted_mask[:len] and leaves the rest 0 (train/model_helpers.py:216-219) --
        # a weight of 0 is never "> 1 + weighted_thresh"
        wp = torch.cat([wp.to(static_sigma.device), torch.zeros(R - wp.shape[0], dtype=wp.dtype, device=static_sigma.device)])
    # (run_args.occl_reg_perc has no effect, here as in the reference: compute_occl_loss ORs an all-ones back mask into the front mask
    # unless use_back is passed, and compute_losses never passes it -- train/model_helpers.py:236-246, 256)
    opts = (use_w, run_args.skewness_val, run_args.entro_mask_thre, run_args.entro_weighted_thresh)
    return _LossTermsFn.apply(static_sigma, temp_sigma, dists, wp, opts)


class _WeightedSqErrFn(torch.autograd.Function):
    """weighted_MSELoss.forward (train/model_helpers.py:284-288) as a HIP kernel under autograd."""

    @staticmethod
    def forward(ctx, preds, gts, weights):
        dt = torch.promote_types(torch.promote_types(preds.dtype, gts.dtype), weights.dtype)
        if dt not in (torch.float32, torch.float64):
            dt = torch.float32
        p, g, w = (t.detach().to(dt).contiguous() for t in (preds, gts, weights))
        out = torch.empty_like(p)
        check(_capi.lib().nca_weighted_sq_err(p.numel(), 1 if dt == torch.float64 else 0, ptr(p), ptr(g), ptr(w), ptr(out), _stream()))
        ctx.save_for_backward(p, g, w)
        ctx.in_dtypes = (preds.dtype, gts.dtype, weights.dtype)
        return out.reshape(preds.shape)

    @staticmethod
    def backward(ctx, g_out):
        p, g, w = ctx.saved_tensors
        go = g_out.detach().to(p.dtype).contiguous()
        need = ctx.needs_input_grad
        outs = [torch.empty_like(p) if n else None for n in need]
        check(_capi.lib().nca_weighted_sq_err_bwd(p.numel(), 1 if p.dtype == torch.float64 else 0, ptr(p), ptr(g), ptr(w), ptr(go),
                                                  ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), _stream()))
        return tuple(None if o is None else o.reshape(g_out.shape).to(dt) for o, dt in zip(outs, ctx.in_dtypes))


def weighted_sq_err(preds, gts, weights):
    _require_cuda(preds, "preds")
    gts, weights = gts.to(preds.device), weights.to(preds.device)
    if not (preds.shape == gts.shape == weights.shape):
        # the reference's expression broadcasts ((preds - gts) ** 2 * weights, train/model_helpers.py:287): expand to the common shape
        # (views: autograd sums the gradient back over the broadcast dimensions)
        try:
            preds, gts, weights = torch.broadcast_tensors(preds, gts, weights)
        except RuntimeError as e:
            raise _capi.NcaError(f"weighted_MSELoss: shapes {tuple(preds.shape)}, {tuple(gts.shape)}, {tuple(weights.shape)} do not broadcast") from e
    return _WeightedSqErrFn.apply(preds, gts, weights)


def fine_depths(sig_s: torch.Tensor, sig_d: Optional[torch.Tensor], z: torch.Tensor, u: torch.Tensor, reduce_max=None) -> torch.Tensor:
    """The sampling half of the hierarchical pass (model_helpers.py:131-148 + sample_pdf): returns the merged, sorted
    depths f32[R, S + n_fine] for coarse fields ``sig_*`` f32[R, S], the shared coarse depths ``z`` f32[S] and the
    uniform draws ``u`` f32[R, n_fine].  The depths are constants of the step here (``fine_depths_autograd`` is the same with the
    reference's backward into the coarse fields).

    ``reduce_max``: for a batch sharded over ranks.  The weights are normalised by the maximum over the WHOLE batch
    (model_helpers.py:139), so the kernel sequence is split: this rank's maximum lands in a device scalar, ``reduce_max``
    (e.g. ``lambda t: dist.all_reduce(t, op=dist.ReduceOp.MAX)``) makes it global in place, and sampling uses the result."""
    _require_cuda(sig_s, "sigma")
    dev = sig_s.device
    R, S = sig_s.shape
    ss = _f32c(sig_s)
    sd = _f32c(sig_d) if sig_d is not None else None
    zz = z.detach().to(device=dev, dtype=torch.float32).contiguous()
    uu = u.detach().to(device=dev, dtype=torch.float32).contiguous()
    if zz.dim() != 1 or zz.shape[0] != S or uu.dim() != 2 or uu.shape[0] != R:
        raise _capi.NcaError("fine_depths takes z[S] shared by all rays and u[R, n_fine]")
    n_fine = uu.shape[1]
    out = torch.empty((R, S + n_fine), dtype=torch.float32, device=dev)
    lib = _capi.lib()
    wbytes = check(lib.nca_fine_depths_workspace(R))
    work = torch.empty(wbytes, dtype=torch.uint8, device=dev)
    if reduce_max is None:
        check(lib.nca_fine_depths(R, S, n_fine, ptr(ss), ptr(sd), ptr(zz), ptr(uu), ptr(out), ptr(work), wbytes, _stream()))
        return out
    wmax = torch.zeros(1, dtype=torch.float32, device=dev)
    check(lib.nca_fine_weight_max(R, S, ptr(ss), ptr(sd), ptr(wmax), ptr(work), wbytes, _stream()))
    reduce_max(wmax)
    check(lib.nca_fine_depths_given_max(R, S, n_fine, ptr(ss), ptr(sd), ptr(zz), ptr(uu), ptr(wmax), ptr(out), _stream()))
    return out


def fine_depths_forward(sig_s, sig_d, z, u, reduce_max=None):
    """The HIP sampler in its two-stage form (so that the maximum it sampled with is at hand): returns
    ``(z_all f32[R, S + n_fine], saved)``; ``saved`` is what ``fine_depths_backward`` needs."""
    dev = sig_s.device
    R, S = sig_s.shape
    ss, sd = _f32c(sig_s), (_f32c(sig_d) if sig_d is not None else None)
    zz = z.detach().to(device=dev, dtype=torch.float32).contiguous()
    uu = u.detach().to(device=dev, dtype=torch.float32).contiguous()
    n_fine = uu.shape[1]
    out = torch.empty((R, S + n_fine), dtype=torch.float32, device=dev)
    lib = _capi.lib()
    wbytes = check(lib.nca_fine_depths_workspace(R))
    work = torch.empty(wbytes, dtype=torch.uint8, device=dev)
    wmax = torch.zeros(1, dtype=torch.float32, device=dev)
    check(lib.nca_fine_weight_max(R, S, ptr(ss), ptr(sd), ptr(wmax), ptr(work), wbytes, _stream()))
    if reduce_max is not None:
        reduce_max(wmax)
    check(lib.nca_fine_depths_given_max(R, S, n_fine, ptr(ss), ptr(sd), ptr(zz), ptr(uu), ptr(wmax), ptr(out), _stream()))
    return out, (ss, sd, zz, uu, wmax)


def fine_depths_backward(saved, g_zall, reduce_sum=None):
    """d loss / d (sigma_s + sigma_d) f32[R, S] of the coarse fields from d loss / d z_all (the reference keeps the sampled
    depths in its autograd graph, model_helpers.py:135-146): nca_fine_depths_bwd / _bwd_max.  ``reduce_sum`` (in-place
    all-reduce SUM) under ray sharding: d loss / d maximum and the number of elements that attain it are sums over the rays
    of ALL ranks."""
    ss, sd, zz, uu, wmax = saved
    dev = ss.device
    R, S = ss.shape
    n_fine = uu.shape[1]
    lib = _capi.lib()
    g = g_zall.detach().to(torch.float32).contiguous()
    g_tot = torch.empty((R, S), dtype=torch.float32, device=dev)
    part = torch.empty((2, R), dtype=torch.float32, device=dev)
    check(lib.nca_fine_depths_bwd(R, S, n_fine, ptr(ss), ptr(sd), ptr(zz), ptr(uu), ptr(wmax), ptr(g), ptr(g_tot), ptr(part[0]), ptr(part[1]), _stream()))
    tot = part.double().sum(1)                         # [d loss / d wmax, number of jumps that attain it] of this rank
    # the leading 1e-10 of every ray's weight vector ties with the maximum only if no jump exceeds it
    tot[1] += float(R) * (wmax[0] == 1e-10).double()
    if reduce_sum is not None:
        reduce_sum(tot)
    each = (tot[0] / tot[1].clamp(min=1.0)).to(torch.float32).reshape(1).contiguous()
    check(lib.nca_fine_depths_bwd_max(R, S, ptr(ss), ptr(sd), ptr(wmax), ptr(each), ptr(g_tot), _stream()))
    return g_tot


class _FineDepthsFn(torch.autograd.Function):
    """z_all = fine_depths(sigma_s, sigma_d; z, u) with the reference's gradient (it leaves the sampled depths in the autograd
    graph, model_helpers.py:135-146): ``fine_depths_forward`` / ``fine_depths_backward``.  ``reduce_max`` as in
    ``fine_depths``; for the backward it must also have a ``sum`` attribute (all-reduce SUM in place)."""

    @staticmethod
    def forward(ctx, sig_s, sig_d, z, u, reduce_max):
        if reduce_max is not None and getattr(reduce_max, "sum", None) is None and (sig_s.requires_grad or (sig_d is not None and sig_d.requires_grad)):
            raise _capi.NcaError("depth gradients under ray sharding need a reducer with a `sum` method (all-reduce SUM): the backward of the "
                                 "batch-wide maximum adds over the ranks")
        out, ctx.keep = fine_depths_forward(sig_s, sig_d, z, u, reduce_max)
        ctx.reduce_max = reduce_max
        ctx.has_d = sig_d is not None
        return out

    @staticmethod
    def backward(ctx, g_zall):
        red = getattr(ctx.reduce_max, "sum", None) if ctx.reduce_max is not None else None
        g_tot = fine_depths_backward(ctx.keep, g_zall, red)
        return g_tot, (g_tot if ctx.has_d else None), None, None, None


def fine_depths_autograd(sig_s, sig_d, z, u, reduce_max=None):
    """``fine_depths`` with the reference's backward into the coarse fields (HIP forward and backward)."""
    _require_cuda(sig_s, "sigma")
    return _FineDepthsFn.apply(sig_s, sig_d, z, u, reduce_max)


def prepare_batch(ids: torch.Tensor, table: torch.Tensor, phases: torch.Tensor, depth: torch.Tensor, t_rand: torch.Tensor,
                  bad_ids: Optional[torch.Tensor] = None):
    """The per-step ray gather (run_composite.py:262-273) and randomize_depth + interval lengths (model_helpers.py:3-12, 73-74) as ONE
    launch: ``ids`` i64[R] into the resident f64 ray table [N,4,3] and its i64 phase vector -> ``(o, d f64[R,3], gt, w f64[R], ph
    i32[R], z f32[S], dists f64[S])``.  Bit-identical to the torch ops it replaces.  An id outside the table never reaches memory:
    it is clamped and counted in ``bad_ids`` (device i32[1], zeroed by the caller; read it when a synchronisation is affordable)."""
    _require_cuda(table, "the ray table")
    dev = table.device
    if table.dtype != torch.float64 or table.dim() != 3 or tuple(table.shape[1:]) != (4, 3) or not table.is_contiguous():
        raise _capi.NcaError("prepare_batch takes the f64 ray table [N, 4, 3]")
    if phases.dtype != torch.int64 or ids.dtype != torch.int64 or not ids.is_contiguous() or not phases.is_contiguous():
        raise _capi.NcaError("prepare_batch takes i64 ray ids and i64 phases")
    if phases.shape[0] != table.shape[0]:
        raise _capi.NcaError("prepare_batch: the phase vector has another length than the ray table")
    if bad_ids is not None and (bad_ids.dtype != torch.int32 or bad_ids.numel() < 1 or bad_ids.device != dev):
        raise _capi.NcaError("prepare_batch: bad_ids is one i32 on the table's device")
    R, S = ids.shape[0], depth.shape[0]
    dep = depth.detach().to(device=dev, dtype=torch.float32).contiguous()
    tr = t_rand.detach().to(device=dev, dtype=torch.float32).contiguous()
    o = torch.empty((R, 3), dtype=torch.float64, device=dev)
    d = torch.empty((R, 3), dtype=torch.float64, device=dev)
    gt = torch.empty(R, dtype=torch.float64, device=dev)
    w = torch.empty(R, dtype=torch.float64, device=dev)
    ph = torch.empty(R, dtype=torch.int32, device=dev)
    z = torch.empty(S, dtype=torch.float32, device=dev)
    dists = torch.empty(S, dtype=torch.float64, device=dev)
    check(_capi.lib().nca_prepare_batch(R, S, ptr(ids), ptr(table), ptr(phases), int(table.shape[0]), ptr(bad_ids), ptr(dep), ptr(tr),
                                        ptr(o), ptr(d), ptr(gt), ptr(w), ptr(ph), ptr(z), ptr(dists), _stream()))
    return o, d, gt, w, ph, z, dists


class BatchSampler:
    """The per-step batch sampler on the device (include/nerfca_hip.h "per-step batch sampling"): the importance sampling of
    run_composite.py:250-260 and the uniform draws of the depth jitter as counter-based Philox streams of (seed, iteration) -- any slot
    range of the global batch can be drawn on its own, and with ``iter_dev`` (device i64[1]) the iteration is read on the device, so a
    captured graph replays consecutive steps without host work."""

    def __init__(self, seed: int, R_global: int, n_var: int, var_ids, non_var_ids, n_rows: int, device):
        self.device = device
        as_dev = lambda a: torch.as_tensor(a, dtype=torch.int64).to(device).contiguous()
        self.var_ids = as_dev(var_ids) if var_ids is not None and len(var_ids) > 0 else None
        self.non_var_ids = as_dev(non_var_ids) if non_var_ids is not None and len(non_var_ids) > 0 else None
        if self.var_ids is None:
            n_var = 0
        self.seed, self.R_global, self.n_var, self.n_rows = int(seed), int(R_global), int(n_var), int(n_rows)

    def desc(self, n_iter: int, iter_dev: Optional[torch.Tensor] = None) -> "_capi.NcaSampler":
        return _capi.NcaSampler(seed=self.seed & ((1 << 64) - 1), n_iter=int(n_iter), iter_dev=ptr(iter_dev), R_global=self.R_global, n_var=self.n_var,
                                var_ids=ptr(self.var_ids), n_var_ids=0 if self.var_ids is None else self.var_ids.numel(),
                                non_var_ids=ptr(self.non_var_ids), n_non_var_ids=0 if self.non_var_ids is None else self.non_var_ids.numel(),
                                n_rows=self.n_rows)

    def ray_ids(self, n_iter: int, slot0: int = 0, count: Optional[int] = None) -> torch.Tensor:
        count = self.R_global - slot0 if count is None else count
        ids = torch.empty(count, dtype=torch.int64, device=self.device)
        d = self.desc(n_iter)
        check(_capi.lib().nca_draw_ray_ids(C.byref(d), slot0, count, ptr(ids), _stream()))
        return ids

    def uniform(self, n_iter: int, n: int, stream_id: int = _capi.RNG_STREAM_JITTER) -> torch.Tensor:
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        d = self.desc(n_iter)
        check(_capi.lib().nca_draw_uniform(C.byref(d), stream_id, n, ptr(out), _stream()))
        return out


def begin_step(sampler: BatchSampler, n_iter: int, slot0: int, R: int, table: torch.Tensor, phases: torch.Tensor, depth: torch.Tensor,
               iter_dev: Optional[torch.Tensor] = None, schedules: Optional["_capi.NcaSchedules"] = None, ids_in: Optional[torch.Tensor] = None,
               t_rand_in: Optional[torch.Tensor] = None, bad_ids: Optional[torch.Tensor] = None, want_draws: bool = False):
    """``prepare_batch`` with everything that changes from step to step made on the device in ONE launch (nca_begin_step): the ids of slots
    ``slot0 .. slot0 + R - 1`` of the global batch (or ``ids_in``), the gather, the jitter draw (or ``t_rand_in``) with the jittered depths and
    interval lengths, and -- ``schedules`` -- the band windows and loss weights of the iteration, written where the descriptor says.  Returns
    ``(o, d, gt, w, ph, z, dists)`` and, with ``want_draws``, also ``ids`` and ``t_rand`` as drawn."""
    _require_cuda(table, "the ray table")
    dev = table.device
    if table.dtype != torch.float64 or table.dim() != 3 or tuple(table.shape[1:]) != (4, 3) or not table.is_contiguous():
        raise _capi.NcaError("begin_step takes the f64 ray table [N, 4, 3]")
    if phases.dtype != torch.int64 or not phases.is_contiguous() or phases.shape[0] != table.shape[0]:
        raise _capi.NcaError("begin_step takes one i64 phase per row of the ray table")
    if ids_in is not None and (ids_in.dtype != torch.int64 or not ids_in.is_contiguous() or ids_in.shape[0] != R or ids_in.device != dev):
        raise _capi.NcaError("ids_in must be a contiguous device i64[R]")
    S = depth.shape[0]
    dep = depth.detach().to(device=dev, dtype=torch.float32).contiguous()
    tr = None if t_rand_in is None else t_rand_in.detach().to(device=dev, dtype=torch.float32).contiguous()
    o = torch.empty((R, 3), dtype=torch.float64, device=dev)
    d = torch.empty((R, 3), dtype=torch.float64, device=dev)
    gt = torch.empty(R, dtype=torch.float64, device=dev)
    w = torch.empty(R, dtype=torch.float64, device=dev)
    ph = torch.empty(R, dtype=torch.int32, device=dev)
    z = torch.empty(S, dtype=torch.float32, device=dev)
    dists = torch.empty(S, dtype=torch.float64, device=dev)
    ids_out = torch.empty(R, dtype=torch.int64, device=dev) if want_draws else None
    t_out = torch.empty(S, dtype=torch.float32, device=dev) if want_draws else None
    sd = sampler.desc(n_iter, iter_dev)
    sd.n_rows = int(table.shape[0])
    check(_capi.lib().nca_begin_step(C.byref(sd), slot0, R, S, C.byref(schedules) if schedules is not None else None, ptr(ids_in), ptr(tr),
                                     ptr(table), ptr(phases), ptr(bad_ids), ptr(dep), ptr(o), ptr(d), ptr(gt), ptr(w), ptr(ph), ptr(z), ptr(dists),
                                     ptr(ids_out), ptr(t_out), _stream()))
    if want_draws:
        return (o, d, gt, w, ph, z, dists), ids_out, t_out
    return o, d, gt, w, ph, z, dists


class FusedAdam:
    """torch.optim.Adam(lr) + LinearLR(1 -> end_factor over total_iters) of run_composite.py:209-215 as ONE library
    launch over the flat parameter buffers of the given models (order as given).  The step counter lives on the
    device, so the launch can be captured in a HIP graph and replayed; ``grads`` are flat f32 tensors per model."""

    def __init__(self, models: Sequence[torch.nn.Module], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, end_factor=1.0, total_iters=0,
                 iter_counter: Optional[torch.Tensor] = None):
        """``iter_counter``: a device i64[1] the kernel increments together with its step count (the training iteration a captured step's
        ``nca_begin_step`` reads)."""
        self.bindings = [m._binding for m in models]
        for b in self.bindings:
            if not b._is_flat():
                b.reflatten()
            _require_cuda(b.flat, "network parameters")
        dev = self.bindings[0].flat.device
        if iter_counter is not None and (iter_counter.dtype != torch.int64 or iter_counter.numel() != 1 or iter_counter.device != dev):
            raise _capi.NcaError("iter_counter must be a device i64[1]")
        self.iter_counter = iter_counter
        self.cfg = _capi.NcaAdam(lr=float(lr), beta1=float(betas[0]), beta2=float(betas[1]), eps=float(eps),
                                 lr_end_factor=float(end_factor), lr_total_iters=int(total_iters), iter_counter=ptr(iter_counter))
        self.exp_avg = [torch.zeros_like(b.flat) for b in self.bindings]
        self.exp_avg_sq = [torch.zeros_like(b.flat) for b in self.bindings]
        self._step = torch.zeros(2, dtype=torch.int64, device=dev)      # [steps taken, the launch's arrival counter (zero between launches)]
        self.step_count = self._step[:1]

    def step(self, grads: Sequence[torch.Tensor]) -> None:
        k = len(self.bindings)
        if len(grads) != k:
            raise ValueError("one flat gradient per model")
        for b, g in zip(self.bindings, grads):
            if g.numel() != b.flat.numel() or g.dtype != torch.float32 or not g.is_contiguous() or g.device != b.flat.device:
                raise _capi.NcaError("gradient must be a contiguous f32 tensor matching the flat parameters")
        arr = lambda ts: (C.c_void_p * k)(*[t.data_ptr() for t in ts])
        n = (C.c_int64 * k)(*[b.flat.numel() for b in self.bindings])
        check(_capi.lib().nca_adam_step(C.byref(self.cfg), k, n, arr([b.flat for b in self.bindings]), arr(grads), arr(self.exp_avg),
                                        arr(self.exp_avg_sq), ptr(self._step), _stream()))
        for b in self.bindings:          # (bias slots of a net without biases stay zero)
            b.zero_gaps(b.flat)


def render_rays(static_model, temp_model, origins, directions, phases, I0, z, dists, act="softplus", single=False, scale=1e-2):
    """Fused query-point -> encoding -> MLP(s) -> activation -> ray sum.

    Composite (``temp_model`` given): returns ``(pix[R], sigma_s[R,S], sigma_d[R,S])`` as
    render_volume_density_composite does (model_helpers.py:72-84).  ``single=True`` renders one net as
    render_volume_density does (un-scaled sigma, model_helpers.py:86-97) and returns ``(pix, sigma)``.
    ``pix`` is f64 when the rays are f64 (the real script), f32 otherwise.
    """
    bs: FieldBinding = static_model._binding
    bd: Optional[FieldBinding] = temp_model._binding if temp_model is not None else None
    if single and bd is not None:
        raise ValueError("single-field render takes exactly one network")
    if bd is not None and bd.prec != bs.prec:
        raise _capi.NcaError("static and dynamic networks must use the same precision (see set_precision)")
    batch = _RayBatch(origins, directions, phases, I0, z, dists, act, single or bd is None, scale)
    params = bs.params() + (bd.params() if bd is not None else [])
    # (inside autograd.Function.forward grad mode is always off and needs_input_grad ignores torch.no_grad(): decide here)
    z_in = z if (torch.is_tensor(z) and z.requires_grad and torch.is_grad_enabled()) else None
    dists_in = dists if (torch.is_tensor(dists) and dists.requires_grad and torch.is_grad_enabled()) else None
    batch.record = torch.is_grad_enabled() and (any(p.requires_grad for p in params) or z_in is not None or dists_in is not None)
    return _RenderFn.apply(batch, bs, bd, len(bs.params()), z_in, dists_in, *params)


def _mlp_forward(binding: "FieldBinding", packed, win, four, prm, N: int, pts, phase) -> torch.Tensor:
    """nca_mlp_fwd, or -- a net on the general kernels -- nca_mlp_fwd_ws with its workspace: raw f32[N, num_output_channels]."""
    lib = _capi.lib()
    c_out = ((binding.net.reserved >> 8) & 0xFF) or 1
    raw = torch.empty((N, c_out), dtype=torch.float32, device=pts.device)
    if _capi.net_is_general(binding.net):
        work, wbytes = _alloc_workspace(lambda cap: check(lib.nca_mlp_fwd_workspace(C.byref(binding.net), binding.prec, N, cap)), pts.device)
        check(lib.nca_mlp_fwd_ws(C.byref(binding.net), binding.prec, ptr(packed), ptr(win), ptr(four), ptr(prm), N,
                                 ptr(pts), ptr(phase), ptr(raw), ptr(work), wbytes, _stream()))
    else:
        check(lib.nca_mlp_fwd(C.byref(binding.net), binding.prec, ptr(packed), ptr(win), ptr(four), ptr(prm), N,
                              ptr(pts), ptr(phase), ptr(raw), _stream()))
    return raw


class _PointsFn(torch.autograd.Function):
    """raw[n] = net(points[n] (, phase[n]))  --  CPPN.forward / Temporal.forward_composite."""

    @staticmethod
    def forward(ctx, binding: FieldBinding, pts: torch.Tensor, phase: Optional[torch.Tensor], *params):
        lib = _capi.lib()
        packed = binding.ensure_packed()
        win, four = binding.module._enc_buffers()
        N = pts.shape[0]
        raw = _mlp_forward(binding, packed, win, four, binding.flat, N, pts, phase)
        ctx.binding, ctx.keep = binding, (packed, win, four, pts, phase)
        return raw

    @staticmethod
    def backward(ctx, g_raw):
        lib = _capi.lib()
        binding = ctx.binding
        packed, win, four, pts, phase = ctx.keep
        N = pts.shape[0]
        g = _f32c(g_raw).reshape(-1)
        grads = torch.empty(binding.flat.numel(), dtype=torch.float32, device=pts.device)
        work, wbytes = _alloc_workspace(lambda cap: check(lib.nca_mlp_bwd_workspace(C.byref(binding.net), binding.prec, N, cap)), pts.device)
        check(lib.nca_mlp_bwd(C.byref(binding.net), binding.prec, ptr(packed), ptr(win), ptr(four), ptr(binding.flat), N,
                              ptr(pts), ptr(phase), ptr(g), ptr(grads), None, ptr(work), wbytes, _stream()))
        binding.last_grad = grads
        return (None, None, None, *binding.split_grads(grads))


def eval_points(model, pts: torch.Tensor, phase: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Evaluate one network on arbitrary points: f32[n, num_input_channels] (, ids[n]) -> f32[n, num_output_channels]."""
    _require_cuda(pts, "query points")
    binding: FieldBinding = model._binding
    c_in, c_out = (binding.net.reserved & 0xFF) or 3, ((binding.net.reserved >> 8) & 0xFF) or 1
    if pts.shape[0] == 0:
        return torch.empty((0, c_out), dtype=torch.float32, device=pts.device)
    p = pts.detach().reshape(-1, c_in).to(torch.float32).contiguous()
    ph = None
    if phase is not None:
        ph = phase.detach().flatten().to(device=p.device, dtype=torch.int32).contiguous()
    return _PointsFn.apply(binding, p, ph, *binding.params())


class _PointsLatentsFn(torch.autograd.Function):
    """raw[n] = net(points[n], table[ids[n]]) with a caller-supplied latent table in place of the module's time_latents.  ``lat`` =
    the latent vector of every point ([N, T]; table[ids[n]] == lat[n]): it is only here so that autograd routes the PER-POINT latent
    gradient the library returns (nca_mlp_bwd's g_latents) to wherever each point's vector came from."""

    @staticmethod
    def forward(ctx, binding: FieldBinding, pts, ids, table, lat, *params):
        lib = _capi.lib()
        packed = binding.ensure_packed()
        win, four = binding.module._enc_buffers()
        N = pts.shape[0]
        prm = binding.flat.detach().clone()                 # natural parameters with the table where the latents sit
        prm[: table.numel()] = table.reshape(-1)
        raw = _mlp_forward(binding, packed, win, four, prm, N, pts, ids)
        ctx.binding, ctx.keep, ctx.n_lat = binding, (packed, win, four, pts, ids, prm), table.numel()
        return raw

    @staticmethod
    def backward(ctx, g_raw):
        lib = _capi.lib()
        binding = ctx.binding
        packed, win, four, pts, ids, prm = ctx.keep
        N = pts.shape[0]
        g = _f32c(g_raw).reshape(-1)
        grads = torch.empty(binding.flat.numel(), dtype=torch.float32, device=pts.device)
        g_lat = torch.empty((N, binding.net.T), dtype=torch.float32, device=pts.device) if ctx.needs_input_grad[4] else None
        work, wbytes = _alloc_workspace(lambda cap: check(lib.nca_mlp_bwd_workspace(C.byref(binding.net), binding.prec, N, cap)), pts.device)
        check(lib.nca_mlp_bwd(C.byref(binding.net), binding.prec, ptr(packed), ptr(win), ptr(four), ptr(prm), N, ptr(pts), ptr(ids), ptr(g), ptr(grads),
                              ptr(g_lat), ptr(work), wbytes, _stream()))
        grads[: ctx.n_lat] = 0.0                            # the table is not the module's time_latents
        return (None, None, None, None, g_lat, *binding.split_grads(grads))


def eval_points_with_latents(model, pts: torch.Tensor, latents: torch.Tensor) -> torch.Tensor:
    """Temporal.query_time: f32[n,3], latent vectors f32[n,T] -> f32[n,1].

    The kernels gather latents from a table by id, so the distinct latent vectors of the call become rows of temporary latent tables
    (P rows per launch).  Gradients with respect to the passed vectors are PER POINT, as the reference's autograd gives them
    (Temporal.py:113-136): the backward returns W0[:, latent columns]^T D_0 of every point (nca_mlp_bwd's g_latents) and autograd adds
    them up wherever rows share a source -- an expanded / indexed smaller table, a leaf with repeated rows, or equal rows that come
    from different tensors alike."""
    _require_cuda(pts, "query points")
    binding: FieldBinding = model._binding
    T, P = binding.net.T, binding.net.P
    want_lat = latents.requires_grad and torch.is_grad_enabled()
    p = pts.detach().reshape(-1, 3).to(torch.float32).contiguous()
    lat_g = latents.reshape(-1, latents.shape[-1]).to(device=p.device, dtype=torch.float32)
    lat = lat_g.detach()
    if lat.shape[0] != p.shape[0] or lat.shape[1] != T:
        raise _capi.NcaError(f"query_time takes one latent vector of {T} values per point")
    if p.shape[0] == 0:
        return torch.empty((0, 1), dtype=torch.float32, device=p.device)
    uniq, inv = torch.unique(lat, dim=0, return_inverse=True)
    out = torch.zeros((p.shape[0], 1), dtype=torch.float32, device=p.device)
    for u0 in range(0, uniq.shape[0], P):                   # one temporary table of P rows per launch
        sel = ((inv >= u0) & (inv < u0 + P)).nonzero().flatten()
        rows = uniq[u0:u0 + P]
        table = torch.cat([rows, torch.zeros((P - rows.shape[0], T), dtype=torch.float32, device=p.device)]) if rows.shape[0] < P else rows
        ids = (inv.index_select(0, sel) - u0).to(torch.int32).contiguous()
        lat_sel = lat_g.index_select(0, sel) if want_lat else lat.index_select(0, sel)
        vals = _PointsLatentsFn.apply(binding, p.index_select(0, sel).contiguous(), ids, table, lat_sel, *binding.params())
        out = out.index_put((sel,), vals)
    return out


class _CompositeFn(torch.autograd.Function):
    """render_volume_density[_composite] on raw fields (model_helpers.py:72-97) as one HIP kernel each way."""

    @staticmethod
    def forward(ctx, raw_s, raw_d, I0, dists, act: int, single: bool, scale: float, f64_out: bool):
        lib = _capi.lib()
        R, S = raw_s.shape
        dev = raw_s.device
        rs = _f32c(raw_s)
        rd = _f32c(raw_d) if raw_d is not None else None
        pix = torch.empty(R, dtype=torch.float64, device=dev)
        sig_s = torch.empty((R, S), dtype=torch.float32, device=dev)
        sig_d = torch.empty((R, S), dtype=torch.float32, device=dev) if not single else None
        i0 = I0.detach().to(device=dev, dtype=torch.float32).expand(R).contiguous()
        dd = dists.detach().to(device=dev, dtype=torch.float64).contiguous()
        check(lib.nca_composite_fwd(R, S, act, 1 if single else 0, scale, ptr(rs), ptr(rd), ptr(i0), ptr(dd), ptr(pix), ptr(sig_s),
                                    ptr(sig_d), _stream()))
        ctx.keep = (rs, rd, dd, act, single, scale)
        if not f64_out:
            pix = pix.to(torch.float32)
        return (pix, sig_s) if single else (pix, sig_s, sig_d)

    @staticmethod
    def backward(ctx, g_pix, g_sig_s, g_sig_d=None):
        lib = _capi.lib()
        rs, rd, dd, act, single, scale = ctx.keep
        R, S = rs.shape
        gp = g_pix.detach().to(torch.float64).contiguous() if g_pix is not None else None
        gs, gd = _f32c(g_sig_s), _f32c(g_sig_d)
        g_rs = torch.empty_like(rs)
        g_rd = torch.empty_like(rd) if rd is not None else None
        check(lib.nca_composite_bwd(R, S, act, 1 if single else 0, scale, ptr(rs), ptr(rd), ptr(dd), ptr(gp), ptr(gs), ptr(gd),
                                    ptr(g_rs), ptr(g_rd), _stream()))
        return g_rs, g_rd, None, None, None, None, None, None


def composite_raw(raw_s, raw_d, I0, dists, act: str, single: bool, scale: float, f64_out: bool):
    """HIP compositing of raw fields [R,S]; returns (pix, sigma_s[, sigma_d])."""
    return _CompositeFn.apply(raw_s, raw_d, I0, dists, act_code(act), single, float(scale), f64_out)
