"""ctypes binding of liblrt_hip.so (the C ABI declared in include/lrt.h).

This is the ONLY compute path of the package: there is no CPU fallback.  If
the shared library is missing or does not load, importing / calling fails
loudly.  torch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
import os, sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LRT_HIP_LIB") or os.path.join(HERE, "csrc", "liblrt_hip.so")   # env override: A/B builds

EXPORTS = ("lrt_abi_version", "lrt_last_error", "lrt_create", "lrt_destroy", "lrt_build", "lrt_build_for_rays", "lrt_build_for_slab", "lrt_forward",
           "lrt_refit", "lrt_backward", "lrt_backward_accum", "lrt_enable_stats", "lrt_get_stats", "lrt_set_option", "lrt_get_option", "lrt_debug_read", "lrt_enable_timing", "lrt_get_timing", "lrt_forward_serial", "lrt_built_count", "lrt_check_forward", "lrt_has_legacy",
           "lrt_status_to_device", "lrt_xchg_msg_words", "lrt_xchg_pack", "lrt_xchg_apply",
           # include/lrt_chamfer.h
           "lrt_chamfer_create", "lrt_chamfer_destroy", "lrt_chamfer_forward", "lrt_chamfer_backward",
           "lrt_chamfer_set_option",
           # include/lrt_knn.h
           "lrt_knn_mean_dist2",
           # include/lrt_preprocess.h
           "lrt_preprocess_forward", "lrt_preprocess_backward")

ABI_VERSION = 4          # LRT_ABI_VERSION of include/lrt.h this binding was written against

_lib = None


class LrtError(RuntimeError):
    pass


def load():
    """Load liblrt_hip.so (after torch, so that both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (must be imported first: brings libamdhip64 into the process)
    if not os.path.exists(LIB_PATH):
        raise LrtError(f"{LIB_PATH} is missing: build it with `python -m lidar_rt_amd.build` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    lib.lrt_abi_version.restype = ci
    lib.lrt_has_legacy.restype = ci
    lib.lrt_last_error.restype = C.c_char_p
    lib.lrt_create.restype = vp; lib.lrt_create.argtypes = [ci]
    lib.lrt_destroy.restype = None; lib.lrt_destroy.argtypes = [vp]
    lib.lrt_build.restype = ci
    lib.lrt_build.argtypes = [vp, ci, vp, vp, vp, vp, cf, vp]
    lib.lrt_refit.restype = ci
    lib.lrt_refit.argtypes = [vp, ci, vp, vp, vp, vp, cf, vp]
    lib.lrt_build_for_rays.restype = ci
    lib.lrt_build_for_rays.argtypes = [vp, ci, vp, vp, vp, vp, cf, ci, vp, vp, vp]
    lib.lrt_build_for_slab.restype = ci
    lib.lrt_build_for_slab.argtypes = [vp, ci, vp, vp, vp, vp, cf, ci, ci, vp, vp, vp]
    lib.lrt_forward.restype = ci
    lib.lrt_forward.argtypes = [vp, ci, ci, vp, vp, ci, ci, ci, vp, vp, ci, vp, vp, vp, vp]
    lib.lrt_backward.restype = ci
    lib.lrt_backward.argtypes = [vp, ci, ci, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp,
                                 vp, vp, vp, vp, vp, vp]
    lib.lrt_backward_accum.restype = ci
    lib.lrt_backward_accum.argtypes = [vp, ci, ci, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp,
                                       vp, vp, vp, vp, vp, vp, vp]
    lib.lrt_enable_stats.restype = ci; lib.lrt_enable_stats.argtypes = [vp, ci]
    lib.lrt_get_stats.restype = ci; lib.lrt_get_stats.argtypes = [vp, C.POINTER(C.c_uint64), vp]
    lib.lrt_enable_timing.restype = ci; lib.lrt_enable_timing.argtypes = [vp, ci]
    lib.lrt_get_timing.restype = ci; lib.lrt_get_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(ci), vp]
    lib.lrt_forward_serial.restype = C.c_longlong; lib.lrt_forward_serial.argtypes = [vp]
    lib.lrt_built_count.restype = ci; lib.lrt_built_count.argtypes = [vp]
    lib.lrt_check_forward.restype = ci; lib.lrt_check_forward.argtypes = [vp, ci]
    lib.lrt_status_to_device.restype = ci; lib.lrt_status_to_device.argtypes = [vp, vp, vp]
    lib.lrt_xchg_msg_words.restype = C.c_longlong; lib.lrt_xchg_msg_words.argtypes = [ci, ci, ci, ci]
    lib.lrt_xchg_pack.restype = ci; lib.lrt_xchg_pack.argtypes = [ci, ci, ci, ci] + [vp] * 8 + [ci, ci, vp]
    lib.lrt_xchg_apply.restype = ci; lib.lrt_xchg_apply.argtypes = [ci, ci, ci, ci, ci, ci, vp, C.c_longlong] + [vp] * 7 + [ci, vp]
    lib.lrt_debug_read.restype = C.c_longlong; lib.lrt_debug_read.argtypes = [vp, ci, vp, C.c_longlong, vp]
    lib.lrt_set_option.restype = ci; lib.lrt_set_option.argtypes = [vp, C.c_char_p, ci]
    lib.lrt_get_option.restype = ci; lib.lrt_get_option.argtypes = [vp, C.c_char_p, C.POINTER(ci)]
    lib.lrt_chamfer_create.restype = vp; lib.lrt_chamfer_create.argtypes = [ci]
    lib.lrt_chamfer_destroy.restype = None; lib.lrt_chamfer_destroy.argtypes = [vp]
    lib.lrt_chamfer_forward.restype = ci
    lib.lrt_chamfer_forward.argtypes = [vp, ci, ci, vp, ci, vp, vp, vp, vp, vp, vp]
    lib.lrt_chamfer_backward.restype = ci
    lib.lrt_chamfer_backward.argtypes = [vp, ci, ci, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.lrt_chamfer_set_option.restype = ci; lib.lrt_chamfer_set_option.argtypes = [vp, C.c_char_p, ci]
    lib.lrt_knn_mean_dist2.restype = ci; lib.lrt_knn_mean_dist2.argtypes = [vp, ci, vp, vp, vp]
    lib.lrt_preprocess_forward.restype = ci; lib.lrt_preprocess_forward.argtypes = [ci, ci, ci] + [vp] * 11
    lib.lrt_preprocess_backward.restype = ci; lib.lrt_preprocess_backward.argtypes = [ci, ci, ci] + [vp] * 14
    if lib.lrt_abi_version() != ABI_VERSION:
        raise LrtError("liblrt_hip.so ABI version mismatch; rebuild with `python -m lidar_rt_amd.build --force`")
    if os.environ.get("LRT_TRACE_CALLS"):                    # developer aid: name every entry point on stderr and wait for the device after it
        lib = _TraceCalls(lib)
    _lib = lib
    return lib


class _TraceCalls:
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, name):
        f = getattr(self._lib, name)
        if not callable(f) or name in ("lrt_last_error", "lrt_abi_version"): return f
        def g(*a):
            sys.stderr.write(f"[lrt] {name}\n"); sys.stderr.flush()
            r = f(*a)
            import torch
            if torch.cuda.is_available(): torch.cuda.synchronize()
            return r
        return g


def has_legacy() -> bool:
    """True when the loaded library is the cross-check build (-DLRT_LEGACY; env LRT_HIP_LIB points at liblrt_hip_legacy.so)."""
    return bool(load().lrt_has_legacy())


def check(rc: int, what: str):
    if rc != 0:
        raise LrtError(f"{what} failed ({rc}): {load().lrt_last_error().decode()}")


def ptr(t):
    """Device pointer of a contiguous torch tensor (None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr()) if t.numel() > 0 else None
