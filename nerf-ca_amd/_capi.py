"""ctypes binding of libnerfca_hip.so (include/nerfca_hip.h).

The library is the product's compute path.  There is no fallback: if it cannot be loaded,
``lib()`` raises, and every op that needs it raises with it.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NERFCA_LIB") or os.path.join(_HERE, "lib", "libnerfca_hip.so")     # (NERFCA_LIB: an experimental build, tools/)

ENC_NONE, ENC_BANDS, ENC_FOURIER = 0, 1, 2
ACT_SIGMOID, ACT_SOFTPLUS, ACT_CLAMP = 0, 1, 2
PREC_F32, PREC_BF16 = 0, 1
ABI_VERSION = 12
OPT_STAGE_FP8 = 1          # (0 is reserved: the retired bf16-staged backward's on-chip threshold)
OPT_RESIDENT_MIN_TILES = 2
OPT_STAGE_FP8_MIN_TILES = 3
OPT_WGRAD_REBUILD_WEIGHT_PCT = 4
OPT_OVERLAP_CUS = 5
OPT_BF16_STORE = 6
STORE_NONE, STORE_F32, STORE_FP8, STORE_BF16, STORE_GENERAL, STORE_KIND_MASK, STORE_SHARED_ENC = 0, 1, 3, 4, 5, 15, 16       # (2 was round 3's bf16-staged store: retired)
WINDOW_NONE, WINDOW_FREE = 0, 1          # NcaWindowSched.kind
RNG_STREAM_IDS, RNG_STREAM_PERM, RNG_STREAM_JITTER, RNG_STREAM_USER = 0, 1, 2, 16
OPT_UNSET = -(1 << 63)     # NCA_OPT_UNSET: "use the process-wide value" in an NcaPlanOpts field
K_PACK, K_FWD, K_BWD_DGRAD, K_BWD_WGRAD, K_BWD_REDUCE, K_LOSS, K_ADAM = 0, 1, 2, 3, 4, 5, 6
KERNEL_KINDS = {"pack": K_PACK, "fwd": K_FWD, "bwd_dgrad": K_BWD_DGRAD, "bwd_wgrad": K_BWD_WGRAD, "bwd_reduce": K_BWD_REDUCE,
                "loss": K_LOSS, "adam": K_ADAM}
TERM_NAMES = ["loss", "pixel", "blendw", "sigma_s_max", "sigma_d_max", "favor_s", "s_entropy", "s_entropy_sum", "d_entropy",
              "d_entropy_sum", "d_occl", "s_l1", "s_l2"]


NET_GENERAL = 0x10000      # NcaNet.reserved: run on the general kernels whatever the width


def net_channels(c_in: int, c_out: int) -> int:
    """NcaNet.reserved for a net with these channel counts (0 = 3 -> 1: the fused kernels' nets)."""
    return (0 if c_in == 3 else c_in) | ((0 if c_out == 1 else c_out) << 8)


def net_is_general(net) -> bool:
    """Does this net run on the general kernels (more than 128 units, other channels than 3 -> 1, or NET_GENERAL set)?"""
    return net.F > 128 or net.reserved != 0


class NcaNet(C.Structure):
    _fields_ = [("F", C.c_int32), ("n_hidden", C.c_int32), ("n_late", C.c_int32), ("enc_mode", C.c_int32),
                ("L", C.c_int32), ("T", C.c_int32), ("P", C.c_int32), ("reserved", C.c_int32)]


class NcaPlanOpts(C.Structure):
    """Per-call planner options (NcaRays.plan_opts): a field other than OPT_UNSET replaces the process-wide tunable for that call."""
    _fields_ = [("stage_fp8", C.c_int64), ("stage_fp8_min_tiles", C.c_int64), ("resident_min_tiles", C.c_int64), ("wgrad_rebuild_weight_pct", C.c_int64),
                ("overlap_cus", C.c_int64), ("bf16_store", C.c_int64)]

    def __init__(self, **kw):
        super().__init__()
        for n, _ in self._fields_:
            setattr(self, n, OPT_UNSET)
        for k, v in kw.items():
            if k not in dict(self._fields_):
                raise NcaError(f"unknown planner option {k!r}")
            setattr(self, k, OPT_UNSET if v is None else int(v))


class NcaRays(C.Structure):
    _fields_ = [("R", C.c_int64), ("S", C.c_int32), ("ray_is_f64", C.c_int32),
                ("origins", C.c_void_p), ("dirs", C.c_void_p),
                ("phase", C.c_void_p), ("phase_stride_r", C.c_int64), ("phase_stride_s", C.c_int64),
                ("z", C.c_void_p), ("z_stride_r", C.c_int64),
                ("dists", C.c_void_p), ("I0", C.c_void_p),
                ("act", C.c_int32), ("single_field", C.c_int32), ("scale", C.c_float), ("store_format", C.c_int32),
                ("plan_opts", C.c_void_p), ("plan_out", C.c_void_p)]          # host pointers: NcaPlanOpts* / NcaPlan* (or NULL)


class NcaLoss(C.Structure):
    _fields_ = [("R", C.c_int64), ("S", C.c_int32), ("use_weighting", C.c_int32), ("skew", C.c_double), ("mask_thre", C.c_double),
                ("weighted_thresh", C.c_double), ("w_favor", C.c_double), ("w_dent", C.c_double), ("w_occl", C.c_double),
                ("w_l1", C.c_double), ("inv_R", C.c_double), ("weights_dev", C.c_void_p), ("unit_mse", C.c_int32), ("reserved", C.c_int32),
                ("g_dists", C.c_void_p), ("dists_work", C.c_void_p), ("term_grads", C.c_void_p),
                # ABI 11 (zero = ABI 10 behaviour): pix formed by the loss kernel from the forward's per-tile ray sums; the terms once more as f32
                ("ray_part", C.c_void_p), ("ray_I0", C.c_void_p), ("pix_out", C.c_void_p), ("ray_nchunk", C.c_int32), ("reserved2", C.c_int32),
                ("terms_f32", C.c_void_p)]


class NcaAdam(C.Structure):
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("lr_end_factor", C.c_double), ("lr_total_iters", C.c_int64), ("iter_counter", C.c_void_p)]


class NcaSampler(C.Structure):
    """The per-step batch sampler on the device (include/nerfca_hip.h: counter-based Philox streams of (seed, iteration))."""
    _fields_ = [("seed", C.c_uint64), ("n_iter", C.c_int64), ("iter_dev", C.c_void_p), ("R_global", C.c_int64), ("n_var", C.c_int64),
                ("var_ids", C.c_void_p), ("n_var_ids", C.c_int64), ("non_var_ids", C.c_void_p), ("n_non_var_ids", C.c_int64), ("n_rows", C.c_int64)]


class NcaWindowSched(C.Structure):
    _fields_ = [("kind", C.c_int32), ("L", C.c_int32), ("window_start", C.c_int32), ("reserved", C.c_int32), ("decay_steps", C.c_int64), ("out", C.c_void_p)]


class NcaWeightSched(C.Structure):
    _fields_ = [("start", C.c_double), ("end", C.c_double), ("steps", C.c_int64), ("delay", C.c_int64)]


class NcaSchedules(C.Structure):
    _fields_ = [("n_windows", C.c_int32), ("reserved", C.c_int32), ("window", NcaWindowSched * 4), ("weight", NcaWeightSched * 4), ("weights_out", C.c_void_p)]


class NcaPlan(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("fwd_store_format", "fwd_launches", "fwd_resident", "bwd_kernel_mode", "bwd_resident", "bwd_launches_per_chunk",
                                         "bwd_onchip", "stage_fp8", "wgrad_jobs", "wgrad_splits", "wgrad_splits_rebuild", "chunks")] + \
               [("wave_tiles", C.c_int64), ("overlap_cus", C.c_int32), ("overlap_forked", C.c_int32), ("reserved", C.c_int64 * 3)]


class NcaError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None

# name -> (restype, argtypes); every symbol include/nerfca_hip.h declares
_P, _I32, _I64 = C.c_void_p, C.c_int32, C.c_int64
SYMBOLS = {
    "nca_abi_version": (C.c_int, []),
    "nca_last_error": (C.c_char_p, []),
    "nca_param_count": (_I64, [C.POINTER(NcaNet)]),
    "nca_packed_bytes": (_I64, [C.POINTER(NcaNet), _I32]),
    "nca_pack_weights": (C.c_int, [C.POINTER(NcaNet), _P, _P, _I32, _P]),
    "nca_pack_weights2": (C.c_int, [C.POINTER(NcaNet), _P, _P, C.POINTER(NcaNet), _P, _P, _I32, _P]),
    "nca_render_fwd_workspace": (_I64, [C.POINTER(NcaRays)]),
    "nca_render_fwd_workspace_nets": (_I64, [C.POINTER(NcaRays), C.POINTER(NcaNet), C.POINTER(NcaNet), _I32, _I64]),
    "nca_render_store_bytes": (_I64, [C.POINTER(NcaRays), C.POINTER(NcaNet), C.POINTER(NcaNet), _I32]),
    "nca_render_fwd": (C.c_int, [C.POINTER(NcaRays), _I32, C.POINTER(NcaNet), _P, _P, _P, C.POINTER(NcaNet), _P, _P, _P, _P,
                                 _P, _P, _P, _P, _I64, _P, _I64, _P]),
    "nca_render_bwd_workspace": (_I64, [C.POINTER(NcaRays), C.POINTER(NcaNet), C.POINTER(NcaNet), _I32, _I64]),
    "nca_render_bwd": (C.c_int, [C.POINTER(NcaRays), _I32, C.POINTER(NcaNet), _P, _P, _P, _P, C.POINTER(NcaNet), _P, _P, _P, _P,
                                 _P, _P, _P, _P, _P, _P, _I64, _P, _I64, _P]),
    "nca_render_bwd_depth": (C.c_int, [C.POINTER(NcaRays), _I32, C.POINTER(NcaNet), _P, _P, _P, _P, C.POINTER(NcaNet), _P, _P, _P, _P,
                                       _P, _P, _P, _P, _P, _P, _P, _I64, _P, _I64, _P]),
    "nca_mlp_fwd": (C.c_int, [C.POINTER(NcaNet), _I32, _P, _P, _P, _P, _I64, _P, _P, _P, _P]),
    "nca_mlp_fwd_workspace": (_I64, [C.POINTER(NcaNet), _I32, _I64, _I64]),
    "nca_mlp_fwd_ws": (C.c_int, [C.POINTER(NcaNet), _I32, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _I64, _P]),
    "nca_mlp_bwd_workspace": (_I64, [C.POINTER(NcaNet), _I32, _I64, _I64]),
    "nca_mlp_bwd": (C.c_int, [C.POINTER(NcaNet), _I32, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "nca_composite_fwd": (C.c_int, [_I64, _I32, _I32, _I32, C.c_float, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nca_composite_bwd": (C.c_int, [_I64, _I32, _I32, _I32, C.c_float, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nca_loss_workspace": (_I64, [_I64]),
    "nca_loss_fwd_bwd": (C.c_int, [C.POINTER(NcaLoss), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "nca_weighted_sq_err": (C.c_int, [_I64, _I32, _P, _P, _P, _P, _P]),
    "nca_weighted_sq_err_bwd": (C.c_int, [_I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nca_fine_depths_workspace": (_I64, [_I64]),
    "nca_fine_depths": (C.c_int, [_I64, _I32, _I32, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "nca_fine_depths_bwd": (C.c_int, [_I64, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nca_fine_depths_bwd_max": (C.c_int, [_I64, _I32, _P, _P, _P, _P, _P, _P]),
    "nca_fine_weight_max": (C.c_int, [_I64, _I32, _P, _P, _P, _P, _I64, _P]),
    "nca_fine_depths_given_max": (C.c_int, [_I64, _I32, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "nca_prepare_batch": (C.c_int, [_I64, _I32, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nca_draw_ray_ids": (C.c_int, [C.POINTER(NcaSampler), _I64, _I64, _P, _P]),
    "nca_draw_uniform": (C.c_int, [C.POINTER(NcaSampler), _I32, _I64, _P, _P]),
    "nca_begin_step": (C.c_int, [C.POINTER(NcaSampler), _I64, _I64, _I32, C.POINTER(NcaSchedules), _P, _P, _P, _P, _P, _P,
                                 _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nca_adam_step": (C.c_int, [C.POINTER(NcaAdam), _I32, C.POINTER(_I64), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P),
                                _P, _P]),
    "nca_get_option": (C.c_int, [_I32, C.POINTER(_I64)]),
    "nca_last_plan": (C.c_int, [C.POINTER(NcaPlan)]),
    "nca_build_info": (C.c_char_p, []),
    "nca_set_option": (C.c_int, [_I32, _I64]),
    "nca_timing_enable": (C.c_int, [_I32]),
    "nca_timing_read": (C.c_int, [_I32, C.POINTER(C.c_double), C.POINTER(_I64)]),
    "nca_timing_reset": (C.c_int, []),
}


def lib() -> C.CDLL:
    """Load (once) and return the HIP library; raise loudly if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NcaError(f"{LIB_PATH} is missing: build it with `make` (or __graft_entry__.build()). "
                           "There is no CPU or PyTorch fallback for the fused ray path.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.nca_abi_version() != ABI_VERSION:
            raise NcaError("libnerfca_hip.so ABI version mismatch")
        _lib = handle
    return _lib


def check(rc: int) -> int:
    """Raise NcaError with the library's message when a call returned a negative code."""
    if rc < 0:
        raise NcaError(f"libnerfca_hip: {lib().nca_last_error().decode()} (code {rc})")
    return rc


def ptr(t) -> Optional[int]:
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def timing_enable(on: bool) -> None:
    check(lib().nca_timing_enable(1 if on else 0))


def timing_reset() -> None:
    check(lib().nca_timing_reset())


def timing_read(kind: str):
    ms, n = C.c_double(0.0), C.c_int64(0)
    check(lib().nca_timing_read(KERNEL_KINDS[kind], C.byref(ms), C.byref(n)))
    return ms.value, n.value


def get_option(opt: int) -> int:
    v = C.c_int64(0)
    check(lib().nca_get_option(opt, C.byref(v)))
    return int(v.value)                         # (a value may be negative: -1 = never / auto)


def set_option(opt: int, value: int) -> None:
    check(lib().nca_set_option(opt, int(value)))


def plan_dict(p: "NcaPlan") -> dict:
    return {n: int(getattr(p, n)) for n, _ in NcaPlan._fields_ if n != "reserved"}


def last_plan() -> dict:
    """What the planner decided in the PROCESS's last nca_render_fwd / nca_render_bwd (see NcaPlan in include/nerfca_hip.h).  A caller
    that shares the process with others reads its own record instead: NcaRays.plan_out (fused.PlanScope / CompositeTrainer.plan)."""
    p = NcaPlan()
    check(lib().nca_last_plan(C.byref(p)))
    return plan_dict(p)


def build_info() -> str:
    return lib().nca_build_info().decode()
