"""ctypes binding of ``libvo_hip.so`` (the C ABI declared in ``include/vo_hip.h``).

The library is built in-tree by ``visual_odometry_amd/csrc/Makefile`` (hipcc, gfx950).  There is no
CPU fallback: if the library is missing, or no MI355X is visible when a device entry point is
called, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os

# torch must be imported first: it loads its own libamdhip64.so.7 and libvo_hip.so has to bind
# to that same runtime instance so that torch's streams and allocations are valid in our launches.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# VO_HIP_LIB: load another build of the same library (kernel ablation experiments only)
LIB_PATH = os.environ.get("VO_HIP_LIB") or os.path.join(_HERE, "libvo_hip.so")

VO_OK = 0
VO_DESC_BAD_VALUE = 1
VO_DESC_BAD_NORM = 2


class VoError(RuntimeError):
    """A libvo_hip.so entry point returned a negative VO_E* code."""


_vp, _i, _d, _f, _sz = C.c_void_p, C.c_int, C.c_double, C.c_float, C.c_size_t

_SIGNATURES = {
    "vo_last_error": (C.c_char_p, []),
    "vo_version": (_i, []),
    "vo_device_check": (_i, [_i]),
    "vo_desc_f32_to_i8": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "vo_knn2_workspace_bytes": (_sz, [_i, _i, _i]),
    "vo_knn2_l2_i8": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _sz, _vp]),
    "vo_ratio_test": (_i, [_vp, _i, _d, _vp, _vp, _vp]),
    "vo_score_fund": (_i, [_vp, _vp, _i, _vp, _i, _f, _vp, _vp]),
    "vo_score_fund_bound": (_i, [_vp, _i, _vp, _i, _f, _vp, _vp]),
    "vo_score_ess": (_i, [_vp, _vp, _i, _vp, _i, _f, _vp, _vp]),
    "vo_score_pnp": (_i, [_vp, _vp, _i, _vp, _vp, C.POINTER(_d), _i, _f, _vp, _vp]),
    "vo_mask_fund": (_i, [_vp, _vp, _i, _f, _vp, _vp]),
    "vo_mask_ess": (_i, [_vp, _vp, _i, _f, _vp, _vp]),
    "vo_mask_pnp": (_i, [_vp, _vp, _vp, C.POINTER(_d), _i, _f, _vp, _vp]),
    "vo_solve_7pt": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp]),
    "vo_solve_5pt": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp]),
    "vo_solve_p3p": (_i, [_vp, _i, _vp, _vp, C.POINTER(_d), _i, _vp, _vp, _vp, _vp]),
    "vo_ransac_replay": (_i, [_vp, _vp, _i, _i, _i, _i, _d, _i, _vp, _vp]),
    "vo_argmax_first": (_i, [_vp, _i, _vp, _vp]),
    "vo_match_compact": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vo_gen_subsets_device": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "vo_triangulate": (_i, [C.POINTER(_d), C.POINTER(_d), _vp, _vp, _i, _vp, _vp]),
    "vo_pose_tail_batch": (_i, [_vp, _vp, _vp, _vp, _vp, C.POINTER(_d), _i, _i, _d, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vo_epnp_batch": (_i, [_vp, _vp, _vp, _vp, _vp, C.POINTER(_d), _i, _i, _vp, _vp, _vp]),
    "vo_lmeds_median_fund": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp]),
    "vo_ransac_workspace_bytes": (_sz, [_i, _i, _i]),
    "vo_ransac_batch": (_i, [_i, _vp, _i, _vp, _vp, C.POINTER(_d), _i, _vp, _i, _f, _d, _i, _i, _vp, _vp, _vp, _vp,
                             _vp, _sz, _vp]),
    "vo_fund_small_iterations": (_i, [_d, _i]),
    "vo_fund_small_workspace_bytes": (_sz, [_d, _i, _i]),
    "vo_fund_small_batch": (_i, [_vp, _i, _vp, _i, _d, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "vo_ransac_layout": (_i, [_i, _i, _i, _vp]),
    "vo_ransac_slice_counts": (_i, [_i, _vp, _i, _i, _i, _vp, _vp, C.POINTER(_d), _i, _vp, _i, _f, _vp, _sz, _vp]),
    "vo_ransac_slice_finish": (_i, [_i, _i, _i, _i, _i, _vp, _i, _d, _i, _vp, _sz, _vp, _vp, _vp, _vp]),
    "vo_ransac_mask_models": (_i, [_i, _vp, _vp, _vp, _vp, C.POINTER(_d), _i, _vp, _i, _f, _vp, _vp]),
    "vo_ransac_static_max_bytes": (_sz, []),
    "vo_ransac_batch_static": (_i, [_i, _vp, _i, _vp, _vp, C.POINTER(_d), _i, _vp, _i, _f, _d, _i, _i, _vp, _vp, _vp, _vp,
                                    _vp]),
    "vo_host_gen_subsets": (_i, [_i, _i, _i, _vp, _vp]),
    "vo_host_gen_subsets_attempts": (_i, [_i, _i, _i, _i, _vp, _vp]),
    "vo_host_update_num_iters": (_i, [_d, _d, _i, _i]),
    "vo_host_epnp": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "vo_synth_sift_like": (_i, [_vp, _i, _i, C.c_uint64, _vp, _vp]),
    "vo_synth_hash64": (_i, [_vp, _i, C.c_long, C.c_uint64, _vp, _vp]),
    "vo_synth_noisy_copies": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, C.c_uint64, _f, _vp, _i, _vp]),
    "vo_profile_enable": (_i, [_i, _i]),
    "vo_profile_collect": (_i, [_vp, _vp, _i]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load() -> C.CDLL:
    """Load libvo_hip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VoError(
                f"{LIB_PATH} not found: build it with `make -C visual_odometry_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        missing = []
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.restype = res
            fn.argtypes = args
        if missing:
            raise VoError(f"{LIB_PATH} lacks symbols declared in include/vo_hip.h: {missing}; rebuild it")
        _lib = lib
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc < 0:
        msg = load().vo_last_error().decode("utf-8", "replace")
        raise VoError(f"{what or 'libvo_hip'} failed with code {rc}: {msg}")


_device_ok = {}


def require_device(device: "int | None" = None) -> None:
    """Fail loudly unless a gfx950 GPU is usable through the loaded library.  device=None checks the
    current device (the one the entry points launch on), never another rank's GPU."""
    if device is None:
        if not torch.cuda.is_available():
            load()
            raise VoError("no HIP device visible to torch: the MI355X path has no CPU fallback")
        device = torch.cuda.current_device()
    if not _device_ok.get(device):
        lib = load()
        if not torch.cuda.is_available():
            raise VoError("no HIP device visible to torch: the MI355X path has no CPU fallback")
        check(lib.vo_device_check(device), "vo_device_check")
        _device_ok[device] = True
