"""Counterpart of the reference's one caller of the operator,
``lib/gaussian_renderer/__init__.py:15-181`` ``raytracing(frame, gaussian_assets, sensor, background, args, ...)``:
same signature, same return dict keys and channel semantics, but the
build2DRectangle + GAS rebuild + trace chain is the fused HIP path
(``Tracer.build_from_gaussians`` -> LBVH, ``Tracer.forward`` -> collect & resolve).

The Gaussian assets / sensor are duck-typed exactly as the reference uses them:
  asset.get_world_xyz(frame) (P,3) | asset.get_opacity (P,1) | asset.get_scaling (P,2)
  asset.get_rotation(frame) -> (actor_quat (1,4) or (4,), local_quat (P,4)) | asset.get_features (P,M,3)
  asset.active_sh_degree
  sensor.get_range_rays(frame) -> (rays_o (H,W,3) [expanded view], rays_d (H,W,3)); sensor.sensor_center[frame]
  or sensor = (rays_o, rays_d, center) tuple (reference :44-46)
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .diff_lidar_tracer import Tracer, TracingSettings

tracer_2dgs = None          # module-level singleton like the reference (:11), created lazily (needs the HIP library)


def quaternion_raw_multiply(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Hamilton product, real part first (lib/utils/general_utils.py:156-174)."""
    aw, ax, ay, az = torch.unbind(a, -1)
    bw, bx, by, bz = torch.unbind(b, -1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw), -1)


# Deferred hit weights (addition; default off = the reference's contract: `accum_gaussian_weight` is complete when raytracing() returns).
# True: a render that a backward will follow (the tracer is in training mode, grad mode on) returns the weights ALL-ZERO and
# loss.backward() fills the same tensor -- train.py reads them after the backward only (:156, :219) -- which takes one float atomic per
# composited hit out of the forward (Tracer(deferred_accum=True), lrt_backward_accum).  Evaluation renders are exact at once either way.
deferred_accum = False
# Bit-reproducible training (addition): Tracer(deterministic=True) -- gradient sums in a fixed order, no learnt state in the forward; implies deferred_accum.
deterministic = False

use_fused_preprocess = True   # module switch: False forces the getter chain of the reference (used by the tests)

# Multi-GPU: set to a lidar_rt_amd.parallel.ShardedTracer and raytracing() traces this rank's azimuth slab only, all-gathers the
# image and, in the backward, exchanges the gradient rows so that EVERY rank holds the full image, the full gradients and the full
# `accum_gaussian_weight` -- train.py's loop (losses, scene.optimize: Adam, densification, pruning) then runs unchanged and identically
# on every rank (lidar_rt_amd.parallel.ShardedTracer: exchange "sparse" or "dense"; not "owner", whose gradients are not replicated).
sharded = None


class _ShardedTrace(torch.autograd.Function):
    """The operator of `_Tracer` (diff_lidar_tracer/__init__.py) across the ranks of a ShardedTracer: forward = local slab + slab
    all_gather, backward = local backward + gradient exchange.  `accum_out` (P,) receives this rank's partial hit weights in the
    forward and the complete ones in the backward (train.py:215-220 reads them after loss.backward())."""

    @staticmethod
    def forward(ctx, st, ray_o, ray_d, means3D, scales, rotations, opacity, shs, deg, bg, accum_out, cull_key=None):
        if st.world > 1 and st.exchange == "owner":
            raise ValueError("renderer.sharded needs a replicated gradient exchange ('sparse', 'dense' or 'auto'), not 'owner'")
        a = [t.detach().contiguous() for t in (means3D, scales, rotations, opacity, shs)]
        out, accum_loc = st.forward(ray_o.contiguous(), ray_d.contiguous(), a[0], a[1], a[2], a[3], a[4], int(deg), bg, cull_key=cull_key)
        accum_out.copy_(accum_loc.reshape(accum_out.shape))
        ctx.st, ctx.deg, ctx.bg, ctx.accum_out = st, int(deg), bg, accum_out
        ctx.fwd = st.last_ctx          # this forward's slab, rays, local output and record serial (a later forward must not replace them)
        ctx.save_for_backward(*a)
        return out

    @staticmethod
    def backward(ctx, dL):
        means, scales, rotations, opacity, shs = ctx.saved_tensors
        g = ctx.st.backward(means, scales, rotations, opacity, shs, ctx.deg, ctx.bg, dL.contiguous(), fwd_ctx=ctx.fwd)
        ctx.accum_out.copy_(g["accum"].reshape(ctx.accum_out.shape))
        # the views belong to a buffer the next step reuses: hand autograd its own copies
        return (None, None, None, g["means"].clone(), g["scales"].clone(), g["rotations"].clone(),
                g["opacities"].reshape(opacity.shape).clone(), g["shs"].clone(), None, None, None, None)


def _fused_inputs(frame, assets, dynamic, decomp):
    """Raw parameters of GaussianModel-like assets (`_xyz`, `_scaling`, `_rotation`, `_opacity`, `bounding_box.frame`)
    through lidar_rt_amd.preprocess.fused_activations; None when an asset does not expose them or when the reference
    would treat its position and its rotation inconsistently (posed means without a composed rotation or vice versa)."""
    from .preprocess import fused_activations, pack_poses
    poses, counts = [], []
    for a, pc in enumerate(assets):
        if not all(hasattr(pc, n) for n in ("_xyz", "_scaling", "_rotation", "_opacity")):
            return None
        bb = getattr(pc, "bounding_box", None)
        fr = bb.frame[frame] if (bb is not None and frame in bb.frame) else None
        composed = dynamic and decomp != "background" and (decomp == "object" or a >= 1)
        if (fr is not None) != composed:
            return None
        poses.append(None if fr is None else (fr[0], fr[1]))
        counts.append(pc._xyz.shape[0])
    if not dynamic and len(assets) > 1:
        return None                       # the reference takes rot_in_local[0] only (gaussian_renderer/__init__.py:115)
    dev = assets[0]._xyz.device
    seg, tab = pack_poses(poses, counts, dev)
    cat = (lambda name: torch.cat([getattr(pc, name) for pc in assets], 0)) if len(assets) > 1 else (lambda name: getattr(assets[0], name))
    return fused_activations(cat("_xyz"), cat("_scaling"), cat("_rotation"), cat("_opacity"), seg, tab)


def raytracing(frame, gaussian_assets, sensor, background, args, scaling_modifier=1.0, override_color=None,
               decomp=False):
    global tracer_2dgs
    if sharded is None:
        if tracer_2dgs is None or tracer_2dgs.deferred_accum != (bool(deferred_accum) or bool(deterministic)) or tracer_2dgs.deterministic != bool(deterministic):
            tracer_2dgs = Tracer(deferred_accum=bool(deferred_accum), deterministic=bool(deterministic))
        # opt.bvh_refit_interval = K > 0: K refits (lrt_refit: same order and topology, new records and boxes) between full LBVH
        # builds while the number of Gaussians is unchanged; results do not depend on it.  0 = rebuild every call (the reference)
        tracer_2dgs.optix_context.refit_interval = int(getattr(getattr(args, "opt", None), "bvh_refit_interval", 0) or 0)
    if decomp == "background":
        gaussian_assets = gaussian_assets[:1]
    elif decomp == "object":
        gaussian_assets = gaussian_assets[1:]
    if isinstance(sensor, tuple):
        rays_o, rays_d, sensor_center = sensor[0], sensor[1], sensor[2]
    elif hasattr(sensor, "get_range_rays"):
        rays_o, rays_d = sensor.get_range_rays(frame)
        sensor_center = sensor.sensor_center[frame]
    else:
        raise ValueError("sensor type not supported")
    if override_color is not None or getattr(getattr(args, "pipe", None), "convert_SHs_python", False) or \
            getattr(getattr(args, "pipe", None), "compute_cov3D_python", False):
        raise NotImplementedError("precomputed colours / covariances are not supported by the tracer kernels "
                                  "(the reference forward ignores them too, forward.cu:261-263)")
    dev = rays_d.device
    e = torch.empty(0, device=dev)
    settings = TracingSettings(None, None, None, None, background.to(dev, torch.float32), 1.0, e, e,
                               gaussian_assets[0].active_sh_degree, sensor_center.to(dev), False, False)
    dynamic = bool(getattr(args, "dynamic", False))
    fused = _fused_inputs(frame, gaussian_assets, dynamic, decomp) if use_fused_preprocess else None
    if fused is not None:
        # one HIP launch for activations + actor transforms + quaternion composition + concatenation (and one for its backward)
        means3D, scales, rotations, opacity = fused
        shs = torch.cat([pc.get_features for pc in gaussian_assets], 0)
    else:
        means, opac, scales, shs, obj_rot, rot_local = [], [], [], [], [], []
        for pc in gaussian_assets:
            means.append(pc.get_world_xyz(frame)); opac.append(pc.get_opacity); scales.append(pc.get_scaling)
            r1, r2 = pc.get_rotation(frame)
            obj_rot.append(r1.expand(r2.shape[0], -1)); rot_local.append(r2)
            shs.append(pc.get_features)
        means3D = torch.cat(means, 0); opacity = torch.cat(opac, 0); scales = torch.cat(scales, 0); shs = torch.cat(shs, 0)
        if decomp == "background" or not dynamic:
            rotations = rot_local[0]
        elif decomp == "object":
            rotations = quaternion_raw_multiply(torch.cat(obj_rot, 0), F.normalize(torch.cat(rot_local, 0), dim=1))
        else:
            rot_obj = quaternion_raw_multiply(torch.cat(obj_rot[1:], 0), F.normalize(torch.cat(rot_local[1:], 0), dim=1))
            rotations = torch.cat([rot_local[0], rot_obj], 0)
    grads3D = torch.zeros_like(means3D, requires_grad=True)
    try:
        means3D.retain_grad()          # train.py:219 reads means3D.grad
    except Exception:
        pass
    if sharded is not None:
        # this rank's azimuth slab + collectives (build, trace and exchange inside ShardedTracer); same outputs on every rank
        accum = torch.zeros(means3D.shape[0], dtype=torch.float32, device=means3D.device)
        rendered = _ShardedTrace.apply(sharded, rays_o, rays_d, means3D, scales, rotations, opacity, shs,
                                       gaussian_assets[0].active_sh_degree, settings.bg, accum,
                                       ("frame", frame, decomp) if isinstance(frame, (int, str)) else None)      # the ray set's name: sizes the culled build (ShardedTracer._cull_sizing)
    else:
        # fused replacement of primitiveCallback(...) + tracer.build_acceleration_structure(vertices, faces, rebuild=True)
        rs_ = frame if isinstance(frame, int) and 0 <= frame < 2 ** 30 else ((hash(frame) & 0x3fffffff) if isinstance(frame, str) else -1)
        if rs_ != getattr(tracer_2dgs, "_ray_set", None):            # the frame names the ray set: the library's learnt per-tile tables are kept per set
            tracer_2dgs.optix_context.set_option("ray_set", rs_); tracer_2dgs._ray_set = rs_
        tracer_2dgs.build_from_gaussians(means3D, scales, rotations, opacity)
        rendered, accum = tracer_2dgs(ray_o=rays_o, ray_d=rays_d, mesh_normals=None, means3D=means3D, grads3D=grads3D,
                                      shs=shs, colors_precomp=None, opacities=opacity, scales=scales,
                                      rotations=rotations, cov3Ds_precomp=None, tracer_settings=settings)
    intensities, rayhit_logits = rendered[:, :, 0:1], rendered[:, :, 1:2]
    raydrop_logits, depth = rendered[:, :, 2:3], rendered[:, :, 3:4]
    if getattr(getattr(args, "opt", None), "use_rayhit", False):
        prob = F.softmax(torch.cat([rayhit_logits, raydrop_logits], dim=-1), dim=-1)
        raydrop_prob = prob[..., 1:2]
    else:
        raydrop_prob = torch.sigmoid(raydrop_logits)
    return {"depth": depth, "intensity": intensities, "raydrop": raydrop_prob, "means3D": means3D,
            "accum_gaussian_weight": accum.unsqueeze(-1)}
