"""MI355X-native drop-in for the reference's ``diff_lidar_tracer`` package
(DLT/diff_lidar_tracer/__init__.py:1-219): ``Tracer`` (nn.Module) and
``TracingSettings`` with the same constructor, method names, argument order,
return tuple and error behaviour.  The native half is ``_C``: the pybind11 torch
extension ``_C_ext`` (csrc/lrt_torch_ext.cpp) on liblrt_hip.so's C ABI, with a
ctypes binding of the same ABI as fallback -- instead of a pybind11/OptiX module.
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _C


class TracingSettings(NamedTuple):
    # field order of DLT/diff_lidar_tracer/__init__.py:139-151
    image_height: Optional[int]      # unused (compatibility)
    image_width: Optional[int]       # unused (compatibility)
    tanfovx: Optional[float]         # unused (compatibility)
    tanfovy: Optional[float]         # unused (compatibility)
    bg: torch.Tensor                 # (3,) background, device tensor
    scale_modifier: float
    viewmatrix: torch.Tensor         # unused by the kernels
    projmatrix: torch.Tensor         # unused by the kernels
    sh_degree: int
    campos: torch.Tensor             # unused by the kernels
    prefiltered: bool
    debug: bool


class _Tracer(torch.autograd.Function):
    """Inputs 5..12 (means3D, grads3D, shs, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)
    receive gradients; ray_o, ray_d and vertices do not (reference :119-134)."""

    @staticmethod
    def forward(ctx, state, training, ray_o, ray_d, vertices, means3D, grads3D, shs, colors_precomp, opacities,
                scales, rotations, cov3Ds_precomp, tracer_settings, check_now=False):
        ts = tracer_settings
        out_f32, out_i32, accum = _C.trace_surfels(
            state, training, ray_o, ray_d, vertices, ts.bg, means3D, shs, ts.sh_degree, colors_precomp, opacities,
            scales, ts.scale_modifier, rotations, cov3Ds_precomp, ts.viewmatrix, ts.projmatrix, ts.campos,
            ts.prefiltered, ts.debug)
        if check_now:
            # decided by Tracer.forward (grad mode is always off in here): no backward will follow to look at the overflow
            # status, so wait for the trace and fail loudly now
            state.check(means3D.device, wait=True)
        ctx.tracer_settings = ts
        ctx.state = state
        # option deferred_accum: a training forward returned `accum` all-zero; the backward of THIS forward writes the sums into the same tensor
        ctx.accum_out = accum if (training and getattr(state, "deferred_accum", False)) else None
        ctx.forward_serial = getattr(state, "last_serial", None)
        ctx.save_for_backward(ray_o, ray_d, vertices, means3D, shs, colors_precomp, opacities, scales, rotations,
                              cov3Ds_precomp, out_f32, out_i32)
        ctx.mark_non_differentiable(accum)
        return out_f32, accum

    @staticmethod
    def backward(ctx, grad_out_f32, _grad_accum):
        ts = ctx.tracer_settings
        (ray_o, ray_d, vertices, means3D, shs, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
         out_f32, out_i32) = ctx.saved_tensors
        (g_means, g_shs, g_colors, g_opac, g_scales, g_rot, g_cov, g_g3) = _C.trace_surfels_backward(
            ctx.state, ray_o, ray_d, vertices, ts.bg, means3D, shs, ts.sh_degree, colors_precomp, opacities, scales,
            ts.scale_modifier, rotations, cov3Ds_precomp, ts.viewmatrix, ts.projmatrix, ts.campos, ts.prefiltered,
            ts.debug, out_f32, out_i32, grad_out_f32, forward_serial=ctx.forward_serial, accum_out=ctx.accum_out)
        g_opac = g_opac.reshape(opacities.shape)
        g_colors = g_colors if colors_precomp.numel() > 0 else None
        g_cov = g_cov if cov3Ds_precomp.numel() > 0 else None
        return (None, None, None, None, None, g_means, g_g3, g_shs, g_colors, g_opac, g_scales, g_rot, g_cov, None, None)


class Tracer(nn.Module):
    def __init__(self, deferred_accum: bool = False, deterministic: bool = False) -> None:
        """deferred_accum (addition; the reference's constructor takes no argument): in training mode with a backward to follow, the
        forward returns `accum` ALL-ZERO and ``loss.backward()`` fills the SAME tensor (the per-Gaussian sums of composite weights,
        forward.cu:268) -- the reference's loop reads them only after the backward (train.py:156,219), and the forward saves one float
        atomic per composited hit.  Evaluation-mode forwards and forwards without a backward stay exact at once.
        deterministic (addition): bit-reproducible results that do not depend on earlier calls -- the gradient sums are taken in a fixed
        order (library option ``deterministic``: a Gaussian's records by ray, the pieces of a run in wave order), the forward keeps no
        learnt state (first-slab widths, carried Morton order, the previous build's box) and the hit weights come from the backward
        (implies deferred_accum: the forward's are float atomics).  Slower (build and forward without their learnt tables, a second pass
        over the records); a resumed training run then repeats the uninterrupted one bit for bit."""
        super().__init__()
        # the reference creates its OptiX context here (zero-argument ctor, module-level singleton in
        # lib/gaussian_renderer/__init__.py:11); ours is a cheap host object, device state is created lazily
        self.optix_context = _C.OptiXStateWrapper("")
        self.vertices = None
        self.deferred_checks = False     # True: forwards without a backward do not wait for the trace; call check() once per batch
        self.deterministic = bool(deterministic)
        self.deferred_accum = bool(deferred_accum) or self.deterministic
        if self.deterministic:
            self.optix_context.set_option("deterministic", 1)
        if self.deferred_accum:
            self.optix_context.deferred_accum = True
            self.optix_context.set_option("deferred_accum", 1)

    # ---- reference surface -------------------------------------------------------------------
    def build_acceleration_structure(self, vertices: torch.Tensor, triangles: torch.Tensor, rebuild: bool = 1):
        self.vertices = vertices
        return _C.build_acceleration_structure(self.optix_context, vertices, triangles, rebuild)

    def forward(self, ray_o: torch.Tensor, ray_d: torch.Tensor, mesh_normals: torch.Tensor, means3D: torch.Tensor,
                grads3D: torch.Tensor, shs: torch.Tensor = None, colors_precomp: torch.Tensor = None,
                opacities: torch.Tensor = None, scales: torch.Tensor = None, rotations: torch.Tensor = None,
                cov3Ds_precomp: torch.Tensor = None, tracer_settings: TracingSettings = None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3Ds_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3Ds_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        dev = means3D.device
        empty = torch.empty(0, dtype=torch.float32, device=dev)
        if shs is None: shs = empty
        if colors_precomp is None: colors_precomp = empty
        if scales is None: scales = empty
        if rotations is None: rotations = empty
        if cov3Ds_precomp is None: cov3Ds_precomp = empty
        vertices = self.vertices if self.vertices is not None else empty
        # The trace is stream-ordered and reports an internal overflow through a status block that the NEXT call into the library
        # reads.  With a backward to follow (training mode, grad mode on, some input requires a gradient) nothing waits here; without
        # one the host waits for the trace and raises now -- unless `deferred_checks` is set (evaluation loops: render many frames,
        # then call `check()` once).
        will_backward = self.training and torch.is_grad_enabled() and any(
            t is not None and t.requires_grad for t in (means3D, grads3D, shs, opacities, scales, rotations))
        check_now = not will_backward and not self.deferred_checks
        return _Tracer.apply(self.optix_context, self.training and will_backward, ray_o, ray_d, vertices, means3D, grads3D, shs,
                             colors_precomp, opacities, scales, rotations, cov3Ds_precomp, tracer_settings, check_now)

    def check(self, device=None):
        """Wait for the most recent forward on `device` and raise if any forward since the last check reported an internal
        overflow (the status bits are sticky).  For loops that run with ``deferred_checks = True``."""
        self.optix_context.check(device, wait=True)

    # ---- additions ---------------------------------------------------------------------------
    def build_from_gaussians(self, means3D, scales, rotations, opacities, scale_modifier: float = 1.0):
        """Fused build2DRectangle + BVH build on the device (no (4P,3) vertex tensor)."""
        return _C.build_from_gaussians(self.optix_context, means3D, scales, rotations, opacities, scale_modifier)


__all__ = ["Tracer", "TracingSettings"]
