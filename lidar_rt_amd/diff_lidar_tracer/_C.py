"""Counterpart of the reference's pybind module ``diff_lidar_tracer._C``
(DLT/ext.cpp:18-23): the same four names, argument orders and return tuples.

Two bindings of the C ABI (include/lrt.h) live behind these names:
  * ``_C_ext`` -- the PyTorch-ROCm C++ extension (csrc/lrt_torch_ext.cpp, pybind11: at::Tensor arguments, contiguity, output
    allocation, current HIP stream and the dead zero outputs in C++), the DEFAULT whenever it has been built;
  * ctypes on liblrt_hip.so (this file) -- the fallback and the binding INTEGRATION.md shows; ``LRT_TORCH_EXT=0`` forces it.
``BACKEND`` names the one in use.  The management calls (options, statistics, timing, status checks) go through ctypes in
both cases: they are not on the per-iteration path.

    OptiXStateWrapper(pkg_dir)                           DLT/optix_tracer/optix_wrapper.cpp:177-233
    build_acceleration_structure(state, vertices, triangles, rebuild)   DLT/trace_surfels.cpp:46-148
    trace_surfels(state, training, ray_o, ray_d, vertices, background, means3D, shs, degree,
                  colors_precomp, opacities, scales, scale_modifier, rotations, transMat_precomp,
                  viewmatrix, projmatrix, campos, prefiltered, debug)
                  -> (out_attr_float32 (H,W,9), out_attr_uint32 (H,W,1) int32 = -1, accum (P,))
                                                                         DLT/trace_surfels.cpp:152-265
    trace_surfels_backward(state, ray_o, ..., debug, out_attr_float32, out_attr_uint32, dL_dout)
                  -> (dL_dmeans3D, dL_dshs, dL_dcolors, dL_dopacities, dL_dscales, dL_drotations,
                      dL_dtransMat_precomp, dL_dgrads3D_abs)              DLT/trace_surfels.cpp:269-386

Difference by design: the acceleration structure is a software LBVH over the
Gaussians' quads, derived on the device from (means, scales, rotations,
opacities) -- exactly the quads ``build2DRectangle`` would produce.  The
``vertices`` / ``triangles`` tensors are validated and remembered but not read
by the kernels; ``build_acceleration_structure`` marks the structure dirty and
``trace_surfels`` (re)builds it from the parameters it is given.
"""
from __future__ import annotations

import os

import torch

from .. import _capi

_ext = None
_ext_error = None
if os.environ.get("LRT_TORCH_EXT", "1") != "0":
    try:
        _capi.load()                       # liblrt_hip.so first (after torch), so that the extension binds the same instance
        from . import _C_ext as _ext       # built in-tree by lidar_rt_amd.build.build_ext
    except Exception as ex:                # not built (or not loadable): the ctypes binding below serves the same surface
        _ext, _ext_error = None, ex
BACKEND = "torch-extension" if _ext is not None else "ctypes"


class _StateAPI:
    """What both state classes offer beyond the reference's opaque handle: options, statistics, timing, status checks
    (management calls through ctypes; `handle(device)` -> (device index, lrt_state* as c_void_p))."""

    def get_option(self, name: str, device=None) -> int:
        """Current value of an option on `device` (hit_cap may have grown after an overflow of the hit record)."""
        import ctypes as C
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        _, h = self.handle(device)
        v = C.c_int(0)
        _capi.check(self._lib.lrt_get_option(h, name.encode(), C.byref(v)), "lrt_get_option")
        return int(v.value)

    def check(self, device=None, wait: bool = True):
        """Raise if the most recent forward on `device` reported an internal overflow (waits for it when `wait`)."""
        want = None
        if device is not None:
            dv = torch.device(device)
            want = dv.index if dv.index is not None else torch.cuda.current_device()
        for idx, h in self._all_handles().items():
            if want is None or idx == want:
                _capi.check(self._lib.lrt_check_forward(h, 1 if wait else 0), "lrt_forward")

    def enable_stats(self, enable: bool = True):
        self.stats_enabled = bool(enable)
        for h in self._all_handles().values():
            self._lib.lrt_enable_stats(h, 1 if enable else 0)

    def built_count(self, device=None) -> int:
        """Primitives in the current LBVH of `device` (fewer than P after a ray-cone culled build)."""
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        idx, h = self.handle(device)
        return int(self._lib.lrt_built_count(h))

    def get_stats(self, device=None):
        import ctypes as C
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        idx, h = self.handle(device)
        arr = (C.c_uint64 * 8)()
        with torch.cuda.device(idx):
            s = torch.cuda.current_stream().cuda_stream
            _capi.check(self._lib.lrt_get_stats(h, arr, C.c_void_p(s)), "lrt_get_stats")
        names = ("candidates", "composited", "passes", "nodes_visited", "prims_tested", "tile_clk_sum", "tile_clk_max",
                 "wave_inserts")
        return {n: int(arr[i]) for i, n in enumerate(names)}

    def enable_timing(self, enable: bool = True):
        self.timing_enabled = bool(enable)
        for h in self._all_handles().values():
            self._lib.lrt_enable_timing(h, 1 if enable else 0)

    def get_timing(self, device=None):
        """HIP-event timings since the last call: {'build'|'fwd'|'bwd'|'colour': (sum_ms, count)} ('colour' lies inside 'fwd')."""
        import ctypes as C
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        idx, h = self.handle(device)
        ms = (C.c_double * 4)(); cnt = (C.c_int * 4)()
        with torch.cuda.device(idx):
            s = torch.cuda.current_stream().cuda_stream
            _capi.check(self._lib.lrt_get_timing(h, ms, cnt, C.c_void_p(s)), "lrt_get_timing")
        return {n: (float(ms[i]), int(cnt[i])) for i, n in enumerate(("build", "fwd", "bwd", "colour"))}



class _CtypesState(_StateAPI):
    """Tracer state (LBVH buffers, workspace) bound to one device, ctypes binding.  The exported name `OptiXStateWrapper` is
    kept for drop-in compatibility; nothing here uses OptiX."""

    def __init__(self, pkg_dir: str = ""):
        self.pkg_dir = pkg_dir
        self._lib = _capi.load()
        self._handles = {}          # device index -> lrt_state*
        self._dirty = {}            # device index -> bool
        self._refit_next = {}       # device index -> bool: build_acceleration_structure(rebuild=0) asked for an update (refit)
        self._built_P = {}
        self._built_mod = {}        # device index -> scale modifier of the current structure
        self.refit_interval = 0     # > 0: that many lrt_refit calls between full builds of an unchanged number of Gaussians
        self._since_full = {}; self._full_P = {}
        self.stats_enabled = False
        self.options = {}

    def _all_handles(self):
        return self._handles

    def handle(self, device: torch.device):
        if device.type != "cuda":
            raise RuntimeError("diff_lidar_tracer: tensors must be on a HIP (cuda) device; there is no CPU path")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        h = self._handles.get(idx)
        if h is None:
            h = self._lib.lrt_create(idx)
            if not h:
                raise _capi.LrtError("lrt_create failed: " + self._lib.lrt_last_error().decode())
            self._handles[idx] = h
            self._dirty[idx] = True
            if self.stats_enabled:
                self._lib.lrt_enable_stats(h, 1)
            if getattr(self, "timing_enabled", False):
                self._lib.lrt_enable_timing(h, 1)
            for k, v in self.options.items():
                _capi.check(self._lib.lrt_set_option(h, k.encode(), int(v)), "lrt_set_option")
        return idx, h

    def mark_dirty(self, refit: bool = False):
        for k in self._dirty:
            self._dirty[k] = True
            self._refit_next[k] = bool(refit)

    def set_option(self, name: str, value: int):
        self.options[name] = int(value)
        for h in self._handles.values():
            _capi.check(self._lib.lrt_set_option(h, name.encode(), int(value)), "lrt_set_option")

    def __del__(self):
        try:
            for h in self._handles.values():
                self._lib.lrt_destroy(h)
            self._handles = {}
        except Exception:
            pass


if _ext is not None:
    class _ExtState(_ext.OptiXStateWrapper, _StateAPI):
        """The extension's state object (C++ owns the per-device lrt_state handles) with the management helpers on top."""

        def __init__(self, pkg_dir: str = ""):
            _ext.OptiXStateWrapper.__init__(self, pkg_dir)
            self._lib = _capi.load()

        def _all_handles(self):
            import ctypes as C
            return {i: C.c_void_p(p) for i, p in self.handles().items()}

        def handle(self, device: torch.device):
            import ctypes as C
            if device.type != "cuda":
                raise RuntimeError("diff_lidar_tracer: tensors must be on a HIP (cuda) device; there is no CPU path")
            idx = device.index if device.index is not None else torch.cuda.current_device()
            return idx, C.c_void_p(self.handle_ptr(idx))

    OptiXStateWrapper = _ExtState
else:
    OptiXStateWrapper = _CtypesState


def _check_f32_cuda(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")           # AT_ASSERTM(is_cuda), trace_surfels.cpp:33-35
    if t.numel() > 0 and t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32")


def _stream_ptr():
    import ctypes as C
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ct_build_acceleration_structure(state, vertices: torch.Tensor, triangles: torch.Tensor,
                                 rebuild: int = 1) -> None:
    # shape checks and messages of DLT/trace_surfels.cpp:53-58
    if vertices.ndimension() != 2 or vertices.size(1) != 3:
        raise RuntimeError("vertices must have dimensions (num_vertices, 3)")
    if triangles.ndimension() != 2 or triangles.size(1) != 3:
        raise RuntimeError("triangles must have dimensions (num_triangles, 3)")
    if not vertices.is_cuda or not triangles.is_cuda:
        raise RuntimeError("vertices/triangles must be CUDA tensors")
    # rebuild == 0 is the reference's OPTIX_BUILD_OPERATION_UPDATE (trace_surfels.cpp:63-73: same topology, new vertex positions): the
    # structure is refitted (lrt_refit) when the next trace finds the number of Gaussians of the last full build, rebuilt otherwise
    state.mark_dirty(refit=(int(rebuild) == 0))
    state.handle(vertices.device)      # create the per-device state eagerly (errors surface here)


def _ct_build_from_gaussians(state, means3D, scales, rotations, opacities, scale_modifier=1.0, cull_rays=None):
    """Fused fast path: LBVH straight from the Gaussian parameters (no vertices tensor).

    cull_rays = (ray_o, ray_d): build only what these rays can reach (lrt_build_for_rays; used by the azimuth-sharded
    tracer, whose ranks see one slab each).  The structure is then valid for THOSE rays only: rebuild before tracing others."""
    for t, n in ((means3D, "means3D"), (scales, "scales"), (rotations, "rotations"), (opacities, "opacities")):
        _check_f32_cuda(t, n)
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    P = means3D.size(0)
    if scales.numel() != 2 * P or rotations.numel() != 4 * P or opacities.numel() != P:
        raise RuntimeError("scales (P,2), rotations (P,4), opacities (P,1) must match means3D (P,3)")
    idx, h = state.handle(means3D.device)
    m, s, r, o = (means3D.detach().contiguous(), scales.detach().contiguous(), rotations.detach().contiguous(),
                  opacities.detach().contiguous())
    with torch.cuda.device(idx):
        if cull_rays is None:
            # refit_interval = K > 0: K refits (same primitive order and tree topology, new records and boxes: ~0.4x the cost)
            # between full builds of an unchanged number of Gaussians; 0 = rebuild every time, like the reference
            since = state._since_full.get(idx)
            can_refit = since is not None and state._full_P.get(idx) == P and P > 0
            asked = state._refit_next.pop(idx, False)
            if can_refit and (asked or (state.refit_interval > 0 and since < state.refit_interval)):
                _capi.check(state._lib.lrt_refit(h, P, _capi.ptr(m), _capi.ptr(s), _capi.ptr(r), _capi.ptr(o),
                                                 float(scale_modifier), _stream_ptr()), "lrt_refit")
                state._since_full[idx] = since + 1
            else:
                _capi.check(state._lib.lrt_build(h, P, _capi.ptr(m), _capi.ptr(s), _capi.ptr(r), _capi.ptr(o),
                                                 float(scale_modifier), _stream_ptr()), "lrt_build")
                state._since_full[idx] = 0; state._full_P[idx] = P
        else:
            state._since_full[idx] = None
            ro, rd = cull_rays
            _check_f32_cuda(ro, "cull ray_o"); _check_f32_cuda(rd, "cull ray_d")
            ro, rd = ro.detach().contiguous(), rd.detach().contiguous()
            if ro.numel() != rd.numel() or rd.numel() % 3:
                raise RuntimeError("cull_rays must be two (...,3) tensors of the same size")
            if rd.ndimension() == 3 and rd.size(2) == 3 and ro.shape == rd.shape:      # a (H, W, 3) slab: the wedge between its edge columns culls too
                _capi.check(state._lib.lrt_build_for_slab(h, P, _capi.ptr(m), _capi.ptr(s), _capi.ptr(r), _capi.ptr(o),
                                                          float(scale_modifier), rd.size(0), rd.size(1), _capi.ptr(ro), _capi.ptr(rd),
                                                          _stream_ptr()), "lrt_build_for_slab")
            else:
                _capi.check(state._lib.lrt_build_for_rays(h, P, _capi.ptr(m), _capi.ptr(s), _capi.ptr(r), _capi.ptr(o),
                                                          float(scale_modifier), rd.numel() // 3, _capi.ptr(ro), _capi.ptr(rd),
                                                          _stream_ptr()), "lrt_build_for_rays")
    state._dirty[idx] = False
    state._built_P[idx] = P
    state._built_mod[idx] = float(scale_modifier)
    # keep the inputs alive until the stream has consumed them (stream-ordered, no sync)
    state._keep = (m, s, r, o)


def _prep(state, ray_o, ray_d, background, means3D, shs, colors_precomp, opacities, scales, rotations,
          transMat_precomp):
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")     # trace_surfels.cpp:178-180
    for t, n in ((ray_o, "ray_o"), (ray_d, "ray_d"), (background, "background"), (means3D, "means3D"),
                 (shs, "shs"), (opacities, "opacities"), (scales, "scales"), (rotations, "rotations")):
        _check_f32_cuda(t, n)
    if colors_precomp is not None and colors_precomp.numel() > 0:
        raise NotImplementedError("colors_precomp is not supported (the reference forward kernel ignores it and "
                                  "reads shs unconditionally, forward.cu:261-263); pass shs")
    if transMat_precomp is not None and transMat_precomp.numel() > 0:
        raise NotImplementedError("cov3Ds_precomp / transMat_precomp is not supported (unused by the reference kernels)")
    if ray_o.ndimension() != 3 or ray_o.size(2) != 3 or ray_d.shape != ray_o.shape:
        raise RuntimeError("ray_o and ray_d must have dimensions (H, W, 3)")
    P = means3D.size(0)
    if P > 0 and (shs.ndimension() != 3 or shs.size(0) != P or shs.size(2) != 3):
        raise RuntimeError("shs must have dimensions (num_points, M, 3)")
    return P


def _ct_trace_surfels(state, training: bool, ray_o, ray_d, vertices, background, means3D, shs,
                  degree: int, colors_precomp, opacities, scales, scale_modifier: float, rotations,
                  transMat_precomp, viewmatrix, projmatrix, campos, prefiltered: bool, debug: bool):
    P = _prep(state, ray_o, ray_d, background, means3D, shs, colors_precomp, opacities, scales, rotations,
              transMat_precomp)
    H, W = ray_o.size(0), ray_o.size(1)
    M = shs.size(1) if shs.numel() > 0 else 0
    dev = means3D.device
    idx, h = state.handle(dev)
    if state._dirty.get(idx, True) or state._built_P.get(idx, -1) != P or state._built_mod.get(idx) != float(scale_modifier):
        _ct_build_from_gaussians(state, means3D, scales, rotations, opacities, scale_modifier)   # also when the scale modifier changed
    ro, rd = ray_o.detach().contiguous(), ray_d.detach().contiguous()      # ray_o is an expanded view upstream
    bg = background.detach().contiguous()
    sh = shs.detach().contiguous()
    out = torch.empty((H, W, 9), dtype=torch.float32, device=dev)
    out_i = torch.empty((H, W, 1), dtype=torch.int32, device=dev)
    accum = torch.empty((P,), dtype=torch.float32, device=dev)
    with torch.cuda.device(idx):
        _capi.check(state._lib.lrt_forward(h, H, W, _capi.ptr(ro), _capi.ptr(rd), P, M, int(degree),
                                           _capi.ptr(sh), _capi.ptr(bg), 1 if training else 0, _capi.ptr(out),
                                           _capi.ptr(out_i), _capi.ptr(accum), _stream_ptr()), "lrt_forward")
    state.last_serial = int(state._lib.lrt_forward_serial(h))
    return out, out_i, accum


def _ct_trace_surfels_backward(state, ray_o, ray_d, vertices, background, means3D, shs,
                           degree: int, colors_precomp, opacities, scales, scale_modifier: float, rotations,
                           transMat_precomp, viewmatrix, projmatrix, campos, prefiltered: bool, debug: bool,
                           out_attr_float32, out_attr_uint32, dL_dout_attr_float32, grads_out=None, forward_serial=None, accum_out=None):
    # accum_out (extension, option deferred_accum): the (P,) accum tensor the forward returned all-zero; this backward writes the sums
    # grads_out (extension): dict of preallocated contiguous fp32 tensors 'means' (P,3), 'shs' (P,M,3), 'opacities' (P,1),
    # 'scales' (P,2), 'rotations' (P,4) to write into (e.g. views of one flat buffer for a fused all-reduce)
    P = _prep(state, ray_o, ray_d, background, means3D, shs, colors_precomp, opacities, scales, rotations,
              transMat_precomp)
    H, W = ray_o.size(0), ray_o.size(1)
    M = shs.size(1) if shs.numel() > 0 else 0
    dev = means3D.device
    idx, h = state.handle(dev)
    if state._built_P.get(idx, -1) != P:
        raise RuntimeError("trace_surfels_backward: acceleration structure does not match (run forward first)")
    if forward_serial is not None and int(state._lib.lrt_forward_serial(h)) != forward_serial:
        # another forward ran on this state since: its hit record replaced ours -> re-trace like the reference
        _capi.check(state._lib.lrt_set_option(h, b"invalidate_record", 1), "lrt_set_option")
    ro, rd = ray_o.detach().contiguous(), ray_d.detach().contiguous()
    bg = background.detach().contiguous()
    m, s, r, o, sh = (means3D.detach().contiguous(), scales.detach().contiguous(), rotations.detach().contiguous(),
                      opacities.detach().contiguous(), shs.detach().contiguous())
    out = out_attr_float32.detach().contiguous()
    dL = dL_dout_attr_float32.detach().contiguous().to(torch.float32)
    opts = dict(dtype=torch.float32, device=dev)
    if grads_out is None:
        # one allocation, [means | shs | opacities | scales | rotations]: the library zero-fills adjacent buffers in one launch
        flat = torch.empty(P * (10 + 3 * M), **opts)
        o0 = 0
        d_means = flat[o0:o0 + 3 * P].view(P, 3); o0 += 3 * P
        d_shs = flat[o0:o0 + 3 * M * P].view(P, M, 3); o0 += 3 * M * P
        d_opac = flat[o0:o0 + P].view(P, 1); o0 += P
        d_scales = flat[o0:o0 + 2 * P].view(P, 2); o0 += 2 * P
        d_rot = flat[o0:o0 + 4 * P].view(P, 4)
    else:
        d_means, d_shs, d_opac = grads_out["means"], grads_out["shs"], grads_out["opacities"]
        d_scales, d_rot = grads_out["scales"], grads_out["rotations"]
        for t_, shp in ((d_means, (P, 3)), (d_shs, (P, M, 3)), (d_opac, (P, 1)), (d_scales, (P, 2)), (d_rot, (P, 4))):
            if tuple(t_.shape) != shp or not t_.is_contiguous() or t_.dtype != torch.float32 or t_.device != dev:
                raise RuntimeError("grads_out tensors must be contiguous float32 device tensors of the gradient shapes")
    if accum_out is not None and (tuple(accum_out.shape) != (P,) or not accum_out.is_contiguous() or accum_out.dtype != torch.float32 or accum_out.device != dev):
        raise RuntimeError("accum_out must be a contiguous float32 device tensor of shape (P,)")
    with torch.cuda.device(idx):
        _capi.check(state._lib.lrt_backward_accum(h, H, W, _capi.ptr(ro), _capi.ptr(rd), P, M, int(degree),
                                                  _capi.ptr(m), _capi.ptr(s), _capi.ptr(r), _capi.ptr(o), _capi.ptr(sh),
                                                  _capi.ptr(bg), _capi.ptr(out), _capi.ptr(dL), _capi.ptr(d_means),
                                                  _capi.ptr(d_shs), _capi.ptr(d_opac), _capi.ptr(d_scales),
                                                  _capi.ptr(d_rot), _capi.ptr(accum_out), _stream_ptr()), "lrt_backward")
    if grads_out is not None:
        return d_means, d_shs, None, d_opac, d_scales, d_rot, None, None
    # dead outputs of the reference (never written by backward.cu): returned as zeros
    d_colors = torch.zeros((P, 3), **opts)
    d_trans = torch.zeros((P, 9), **opts)
    d_g3abs = torch.zeros((P, 3), **opts)
    return d_means, d_shs, d_colors, d_opac, d_scales, d_rot, d_trans, d_g3abs


# ---- the four names of DLT/ext.cpp:18-23 (+ build_from_gaussians): the extension for its own state objects, ctypes otherwise
def _is_ext(state) -> bool:
    return _ext is not None and isinstance(state, _ext.OptiXStateWrapper)


def build_acceleration_structure(state, vertices: torch.Tensor, triangles: torch.Tensor, rebuild: int = 1) -> None:
    return (_ext.build_acceleration_structure if _is_ext(state) else _ct_build_acceleration_structure)(state, vertices, triangles, rebuild)


def build_from_gaussians(state, means3D, scales, rotations, opacities, scale_modifier=1.0, cull_rays=None):
    return (_ext.build_from_gaussians if _is_ext(state) else _ct_build_from_gaussians)(state, means3D, scales, rotations, opacities,
                                                                                      scale_modifier, cull_rays=cull_rays)


def trace_surfels(state, training: bool, ray_o, ray_d, vertices, background, means3D, shs, degree: int, colors_precomp, opacities,
                  scales, scale_modifier: float, rotations, transMat_precomp, viewmatrix, projmatrix, campos, prefiltered: bool, debug: bool):
    return (_ext.trace_surfels if _is_ext(state) else _ct_trace_surfels)(state, training, ray_o, ray_d, vertices, background, means3D, shs, degree,
                                                                         colors_precomp, opacities, scales, scale_modifier, rotations, transMat_precomp,
                                                                         viewmatrix, projmatrix, campos, prefiltered, debug)


def trace_surfels_backward(state, ray_o, ray_d, vertices, background, means3D, shs, degree: int, colors_precomp, opacities, scales,
                           scale_modifier: float, rotations, transMat_precomp, viewmatrix, projmatrix, campos, prefiltered: bool, debug: bool,
                           out_attr_float32, out_attr_uint32, dL_dout_attr_float32, grads_out=None, forward_serial=None, accum_out=None):
    return (_ext.trace_surfels_backward if _is_ext(state) else _ct_trace_surfels_backward)(
        state, ray_o, ray_d, vertices, background, means3D, shs, degree, colors_precomp, opacities, scales, scale_modifier, rotations,
        transMat_precomp, viewmatrix, projmatrix, campos, prefiltered, debug, out_attr_float32, out_attr_uint32, dL_dout_attr_float32,
        grads_out=grads_out, forward_serial=forward_serial, accum_out=accum_out)
