"""Training-loop counterpart: what zju3dv/LiDAR-RT's ``train.py`` needs around the tracer,
free of tensorflow / open3d / plyfile, on the MI355X operators of this package.

    GaussianAsset      <-> lib/scene/gaussian_model.py: GaussianModel (parameter store, activations, Adam groups,
                           densification statistics, clone / split / prune, opacity reset, capture() / restore())
    GaussianScene      <-> lib/dataloader/gs_loader.py:220-298 SceneLidar (training_setup, optimize(), save / restore)
    RangeFrames        <-> the slice of lib/scene/lidar_sensor.py that the loop reads (range image store, rays, masks)
    training_step      <-> train.py:125-220 (render, the five losses, backward, optimize)

The on-disk checkpoint keeps the reference's layout: ``torch.save((list of per-asset 12-tuples, iteration), path)`` with
``(active_sh_degree, xyz, features_dc, features_rest, scaling, rotation, opacity, max_radii2D, xyz_gradient_accum,
denom, optimizer.state_dict(), spatial_lr_scale)`` (gaussian_model.py:58-72), so checkpoints move between the two.

Everything here is bookkeeping in PyTorch (it is in the reference too); the device work goes through
``renderer.raytracing`` (fused pre-processing + tracer), ``chamfer3D`` and ``simple_knn`` of this package.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import nn

SH_C0 = 0.28209479177387814


def inverse_sigmoid(x: torch.Tensor) -> torch.Tensor:
    return torch.log(x / (1 - x))


def default_options() -> SimpleNamespace:
    """The hyper-parameters the loop reads, with the reference's values (configs/base.yaml:14-25, configs/exp.yaml:20-43)."""
    return SimpleNamespace(
        iterations=30_000, position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01,
        position_lr_max_steps=30_000, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001,
        densify_from_iter=500, densify_until_iter=15_000, densification_interval=100, opacity_reset_interval=3000,
        densify_scale_threshold=0.0002, densify_grad_threshold=0.0002, densify_weight_threshold=0.0,
        prune_size_threshold=0.1, thresh_opa_prune=0.003, lambda_cd=0.01, lambda_depth_l1=0.1, lambda_intensity_l1=0.85,
        lambda_intensity_l2=0.0, lambda_intensity_dssim=0.15, lambda_raydrop_bce=0.01, lambda_reg=0.01, use_rayhit=False,
        bvh_refit_interval=0)   # not in the reference: K refits between full LBVH builds (renderer.raytracing), 0 = rebuild per call


def expon_lr(step: int, lr_init: float, lr_final: float, delay_mult: float = 1.0, delay_steps: int = 0,
             max_steps: int = 1_000_000) -> float:
    """Log-linear interpolation lr_init -> lr_final with an optional eased start (general_utils.py:30-64)."""
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    rate = 1.0
    if delay_steps > 0:
        rate = delay_mult + (1 - delay_mult) * math.sin(0.5 * math.pi * min(max(step / delay_steps, 0.0), 1.0))
    t = min(max(step / max_steps, 0.0), 1.0)
    return rate * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


class GaussianAsset:
    """One set of 2-D Gaussians (the background, or one actor with a ``bounding_box``)."""

    GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")       # Adam group names, gaussian_model.py:193-200

    def __init__(self, max_sh_degree: int = 3, extent: float = 1.0, bounding_box=None, dimension: int = 2):
        self.max_sh_degree, self.active_sh_degree = max_sh_degree, 0
        self.extent, self.spatial_lr_scale = float(extent), float(extent)
        self.bounding_box, self.dimension = bounding_box, dimension
        self._xyz = self._features_dc = self._features_rest = self._scaling = self._rotation = self._opacity = None
        self.max_radii2D = self.xyz_gradient_accum = self.denom = None
        self.optimizer: Optional[torch.optim.Adam] = None
        self._lr_args = None

    # ---- construction ---------------------------------------------------------------------------------------------
    @classmethod
    def from_points(cls, points: torch.Tensor, intensity: torch.Tensor, normals: Optional[torch.Tensor] = None, **kw):
        """Initialisation from a point cloud (gaussian_model.py:155-184): scales from the mean squared distance to the
        3 nearest neighbours (``distCUDA2``), opacity 0.1, DC feature = RGB2SH(colour), flat quaternions."""
        from .simple_knn._C import distCUDA2          # the package's own operator, not whatever top-level `simple_knn` is installed
        self = cls(**kw)
        pts = points.float().contiguous()
        P, dev = pts.shape[0], pts.device
        feats = torch.zeros((P, 3, (self.max_sh_degree + 1) ** 2), device=dev)
        feats[:, :3, 0] = (intensity.float().reshape(P, -1).expand(P, 3) - 0.5) / SH_C0
        dist2 = torch.clamp_min(distCUDA2(pts), 1e-7)
        scales = torch.log(torch.sqrt(dist2))[:, None].repeat(1, self.dimension)
        rots = torch.rand((P, 4), device=dev) if normals is None else _quaternions_with_normal(normals.float())
        self._set(pts, feats[:, :, 0:1].transpose(1, 2).contiguous(), feats[:, :, 1:].transpose(1, 2).contiguous(),
                  scales, rots, inverse_sigmoid(0.1 * torch.ones((P, 1), device=dev)))
        return self

    @classmethod
    def from_tensors(cls, xyz, features_dc, features_rest, scaling, rotation, opacity, **kw):
        self = cls(**kw)
        self._set(xyz, features_dc, features_rest, scaling, rotation, opacity)
        return self

    def _set(self, xyz, f_dc, f_rest, scaling, rotation, opacity):
        mk = lambda t: nn.Parameter(t.detach().clone().float().contiguous().requires_grad_(True))
        self._xyz, self._features_dc, self._features_rest = mk(xyz), mk(f_dc), mk(f_rest)
        self._scaling, self._rotation, self._opacity = mk(scaling), mk(rotation), mk(opacity)
        self.max_radii2D = torch.zeros(self._xyz.shape[0], device=self._xyz.device)

    # ---- the getters raytracing() falls back to (gaussian_model.py:112-148) ---------------------------------------------
    get_scaling = property(lambda s: torch.exp(s._scaling))
    get_opacity = property(lambda s: torch.sigmoid(s._opacity))
    get_local_xyz = property(lambda s: s._xyz)
    get_features = property(lambda s: torch.cat((s._features_dc, s._features_rest), dim=1))

    def _pose(self, ts):
        bb = self.bounding_box
        return bb.frame[ts] if (bb is not None and ts in bb.frame) else None

    def get_rotation(self, ts=0.0):
        ps = self._pose(ts)
        obj = ps[1] if ps is not None else torch.zeros((1, 4), device=self._rotation.device)
        return obj, F.normalize(self._rotation)

    def get_world_xyz(self, ts=0.0):
        ps = self._pose(ts)
        if ps is None:
            return self._xyz
        return self._xyz @ _rotation_matrix(ps[1].reshape(1, 4)).squeeze(0).T + ps[0]

    def oneupSHdegree(self):
        self.active_sh_degree = min(self.active_sh_degree + 1, self.max_sh_degree)

    # ---- optimiser ------------------------------------------------------------------------------------------------
    def _params(self) -> Dict[str, nn.Parameter]:
        return {"xyz": self._xyz, "f_dc": self._features_dc, "f_rest": self._features_rest, "opacity": self._opacity,
                "scaling": self._scaling, "rotation": self._rotation}

    def training_setup(self, opt):
        P, dev = self._xyz.shape[0], self._xyz.device
        self.densify_scale_threshold, self.densify_weight_threshold = opt.densify_scale_threshold, opt.densify_weight_threshold
        self.xyz_gradient_accum, self.denom = torch.zeros((P, 1), device=dev), torch.zeros((P, 1), device=dev)
        lrs = {"xyz": opt.position_lr_init * self.spatial_lr_scale, "f_dc": opt.feature_lr, "f_rest": opt.feature_lr / 20.0,
               "opacity": opt.opacity_lr, "scaling": opt.scaling_lr, "rotation": opt.rotation_lr}
        pr = self._params()
        # same update rule as the reference's torch.optim.Adam(lr=0.0, eps=1e-15); on the GPU the fused implementation runs the
        # six groups in one multi-tensor kernel (1.45 -> 0.35 ms per step at 1 M Gaussians)
        self.optimizer = torch.optim.Adam([{"params": [pr[n]], "lr": lrs[n], "name": n} for n in self.GROUPS], lr=0.0, eps=1e-15,
                                          fused=bool(dev.type == "cuda"))
        self._lr_args = dict(lr_init=opt.position_lr_init * self.spatial_lr_scale, lr_final=opt.position_lr_final * self.spatial_lr_scale,
                             delay_mult=opt.position_lr_delay_mult, max_steps=opt.position_lr_max_steps)

    def update_learning_rate(self, iteration: int) -> float:
        lr = expon_lr(iteration, **self._lr_args)
        for g in self.optimizer.param_groups:
            if g["name"] == "xyz":
                g["lr"] = lr
        return lr

    def _rewrite(self, new_value, keep_state, only=None):
        """Replace every optimised tensor (or only the group `only`) by ``new_value(name, old)`` and its Adam moments by
        ``keep_state(name, moment)`` (the optimiser surgery behind pruning, densification and the opacity reset,
        gaussian_model.py:227-289).  A replaced tensor is a NEW Parameter without a gradient: the optimiser step that follows in the
        same iteration skips it, exactly like the reference's."""
        out = dict(self._params())
        for g in self.optimizer.param_groups:
            if only is not None and g["name"] != only:
                continue
            old = g["params"][0]
            st = self.optimizer.state.pop(old, None)
            new = nn.Parameter(new_value(g["name"], old.detach()).contiguous().requires_grad_(True))
            if st is not None:
                st["exp_avg"], st["exp_avg_sq"] = keep_state(g["name"], st["exp_avg"]), keep_state(g["name"], st["exp_avg_sq"])
                self.optimizer.state[new] = st
            g["params"][0] = new
            out[g["name"]] = new
        self._xyz, self._features_dc, self._features_rest = out["xyz"], out["f_dc"], out["f_rest"]
        self._opacity, self._scaling, self._rotation = out["opacity"], out["scaling"], out["rotation"]

    def prune_points(self, mask: torch.Tensor):
        keep = ~mask
        self._rewrite(lambda n, t: t[keep], lambda n, m: m[keep])
        self.xyz_gradient_accum, self.denom, self.max_radii2D = self.xyz_gradient_accum[keep], self.denom[keep], self.max_radii2D[keep]

    def _append(self, new: Dict[str, torch.Tensor]):
        self._rewrite(lambda n, t: torch.cat((t, new[n]), 0), lambda n, m: torch.cat((m, torch.zeros_like(new[n])), 0))
        P, dev = self._xyz.shape[0], self._xyz.device
        self.xyz_gradient_accum, self.denom = torch.zeros((P, 1), device=dev), torch.zeros((P, 1), device=dev)
        self.max_radii2D = torch.zeros(P, device=dev)

    def reset_opacity(self):
        """Opacities are pulled down to <= 0.01 and their Adam moments restart (gaussian_model.py:216-219)."""
        new = inverse_sigmoid(torch.min(self.get_opacity, torch.ones_like(self.get_opacity) * 0.01)).detach()
        # only the opacity group is replaced (replace_tensor_to_optimizer): the other five keep their gradients and are stepped
        self._rewrite(lambda n, t: new, lambda n, m: torch.zeros_like(m), only="opacity")

    # ---- densification --------------------------------------------------------------------------------------------
    def add_densification_stats(self, mean_grads: torch.Tensor, update_filter: torch.Tensor):
        self.xyz_gradient_accum += torch.norm(mean_grads, dim=-1, keepdim=True)
        self.denom += update_filter.reshape(-1, 1).to(self.denom.dtype)       # == denom[update_filter] += 1, without nonzero()

    def _select(self, mask):
        return {n: t.detach()[mask] for n, t in self._params().items()}

    def densify_and_prune(self, opt, size_limit: Optional[float]) -> Tuple[int, int, int, int]:
        """Clone small / split large Gaussians whose mean positional gradient is above the threshold, then prune
        transparent, oversized and (for actors) out-of-box ones (gaussian_model.py:311-411)."""
        grads = (self.xyz_gradient_accum / self.denom).nan_to_num(0.0).squeeze(-1)
        big = torch.max(self.get_scaling, dim=1).values > self.densify_scale_threshold * self.extent
        # clone: copies of the selected Gaussians are appended unchanged
        sel = (grads >= opt.densify_grad_threshold) & ~big
        n_clone = int(sel.sum())
        self._append(self._select(sel))
        # split: two samples of each selected Gaussian's own distribution, 1.6x smaller, replace it
        P0 = self._xyz.shape[0]
        g2 = torch.zeros(P0, device=grads.device); g2[:grads.shape[0]] = grads
        sel = (g2 >= opt.densify_grad_threshold) & (torch.max(self.get_scaling, dim=1).values > self.densify_scale_threshold * self.extent)
        n_split, N = int(sel.sum()), 2
        chosen = self._select(sel)
        stds = self.get_scaling.detach()[sel].repeat(N, 1)
        stds3 = torch.cat([stds, torch.zeros_like(stds[:, :1])], -1) if self.dimension == 2 else stds
        offs = torch.bmm(_rotation_matrix(chosen["rotation"]).repeat(N, 1, 1), torch.normal(torch.zeros_like(stds3), stds3).unsqueeze(-1)).squeeze(-1)
        new = {n: t.repeat(N, *([1] * (t.dim() - 1))) for n, t in chosen.items()}
        new["xyz"] = offs + chosen["xyz"].repeat(N, 1)
        new["scaling"] = torch.log(stds / (0.8 * N))
        self._append(new)
        self.prune_points(torch.cat((sel, torch.zeros(N * n_split, dtype=torch.bool, device=sel.device))))
        # prune
        low = (self.get_opacity < opt.thresh_opa_prune).squeeze(-1)
        n_opa, n_scale, mask = int(low.sum()), 0, low
        if size_limit:
            huge = self.get_scaling.max(dim=1).values > 0.1 * self.extent * opt.prune_size_threshold
            n_scale, mask = int(huge.sum()), low | huge
            if self.bounding_box is not None and self._xyz.shape[0] > 0:
                mask = mask | ~self._inside_box(2)
        if int(mask.sum()) < self._xyz.shape[0]:
            self.prune_points(mask)
        return n_clone, n_split, n_scale, n_opa

    def _inside_box(self, n_samples: int) -> torch.Tensor:
        """Gaussians whose random samples all fall into the actor's tracking box (gaussian_model.py:381-404)."""
        s = self.get_scaling.detach()
        s3 = torch.cat([s, torch.zeros_like(s[:, :1])], -1) if self.dimension == 2 else s
        smp = torch.normal(torch.zeros_like(s3)[:, None].expand(-1, n_samples, -1), s3[:, None].expand(-1, n_samples, -1))
        R = _rotation_matrix(self._rotation.detach())[:, None].expand(-1, n_samples, -1, -1)
        pts = torch.matmul(R, smp.unsqueeze(-1)).squeeze(-1) + self._xyz.detach()[:, None]
        P = pts.shape[0]
        return torch.all((pts >= self.bounding_box.min_xyz).view(P, -1), -1) & torch.all((pts <= self.bounding_box.max_xyz).view(P, -1), -1)

    def box_reg_loss(self):
        if self.bounding_box is None:
            return 0
        over = torch.clamp_min(self._xyz - self.bounding_box.max_xyz, 0.).mean() + torch.clamp_min(self.bounding_box.min_xyz - self._xyz, 0.).mean()
        return over / self.extent * 100 + (self.get_scaling.max(dim=1).values / self.extent).mean()

    # ---- checkpoint (the reference's tuple, gaussian_model.py:58-106) ---------------------------------------------
    def capture(self):
        return (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation,
                self._opacity, self.max_radii2D, self.xyz_gradient_accum, self.denom, self.optimizer.state_dict(),
                self.spatial_lr_scale)

    def restore(self, model_args, opt):
        (self.active_sh_degree, xyz, f_dc, f_rest, scaling, rotation, opacity, self.max_radii2D, grad_accum, denom,
         opt_dict, self.spatial_lr_scale) = model_args
        mk = lambda t: t if isinstance(t, nn.Parameter) else nn.Parameter(t.requires_grad_(True))
        self._xyz, self._features_dc, self._features_rest = mk(xyz), mk(f_dc), mk(f_rest)
        self._scaling, self._rotation, self._opacity = mk(scaling), mk(rotation), mk(opacity)
        self.training_setup(opt)
        self.xyz_gradient_accum, self.denom = grad_accum, denom
        self.optimizer.load_state_dict(opt_dict)


def _rotation_matrix(q: torch.Tensor) -> torch.Tensor:
    """(N,4) quaternions (w,x,y,z), normalised here -> (N,3,3) (general_utils.py:176-197)."""
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).view(-1, 3, 3)


def _quaternions_with_normal(n: torch.Tensor) -> torch.Tensor:
    """Quaternions whose third rotation axis is the given normal, with a random spin about it."""
    n = F.normalize(n, dim=1)
    helper = torch.where((n[:, :1].abs() < 0.9).expand(-1, 3), torch.tensor([1.0, 0, 0], device=n.device).expand_as(n),
                         torch.tensor([0, 1.0, 0], device=n.device).expand_as(n))
    a = F.normalize(torch.cross(helper, n, dim=1), dim=1)
    b = torch.cross(n, a, dim=1)
    ang = torch.rand(n.shape[0], device=n.device) * 2 * math.pi
    a2 = a * torch.cos(ang)[:, None] + b * torch.sin(ang)[:, None]
    b2 = torch.cross(n, a2, dim=1)
    R = torch.stack([a2, b2, n], -1)                                   # columns = axes
    return F.normalize(_matrix_to_quaternion(R), dim=1)


def _matrix_to_quaternion(R: torch.Tensor) -> torch.Tensor:
    """(N,3,3) rotations -> (N,4) (w,x,y,z); picks the numerically largest component first (stable near 180 degrees)."""
    m00, m11, m22 = R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]
    q2 = torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], -1)   # 4 w^2, 4 x^2, ...
    cand = torch.stack([
        torch.stack([q2[:, 0], R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], -1),
        torch.stack([R[:, 2, 1] - R[:, 1, 2], q2[:, 1], R[:, 1, 0] + R[:, 0, 1], R[:, 0, 2] + R[:, 2, 0]], -1),
        torch.stack([R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] + R[:, 0, 1], q2[:, 2], R[:, 2, 1] + R[:, 1, 2]], -1),
        torch.stack([R[:, 1, 0] - R[:, 0, 1], R[:, 0, 2] + R[:, 2, 0], R[:, 2, 1] + R[:, 1, 2], q2[:, 3]], -1)], 1)    # (N,4,4)
    best = q2.argmax(1)
    q = cand[torch.arange(R.shape[0], device=R.device), best]
    q = q / (2 * torch.sqrt(q2.gather(1, best[:, None]).clamp_min(1e-12)))
    return torch.where(q[:, :1] < 0, -q, q)


class GaussianScene:
    """The per-asset loop of SceneLidar (gs_loader.py:220-298): asset 0 is the background, the others are actors."""

    def __init__(self, assets: Sequence[GaussianAsset]):
        self.gaussians_assets: List[GaussianAsset] = list(assets)

    def training_setup(self, opt):
        for g in self.gaussians_assets:
            g.training_setup(opt)

    def update_learning_rate(self, iteration):
        for g in self.gaussians_assets:
            g.update_learning_rate(iteration)

    def oneupSHdegree(self):
        for g in self.gaussians_assets:
            g.oneupSHdegree()

    def save(self, iteration: int, path: str):
        torch.save(([g.capture() for g in self.gaussians_assets], iteration), path)

    def restore(self, model_params, opt):
        for g, mp in zip(self.gaussians_assets, model_params):
            g.restore(mp, opt)

    def optimize(self, opt, iteration: int, mean_grads: torch.Tensor, accum_weights: torch.Tensor):
        """Densification statistics / densify & prune / opacity reset / Adam step per asset (gs_loader.py:243-298)."""
        tot = [0, 0, 0, 0]
        begin = 0
        for g in self.gaussians_assets:
            n = g._xyz.shape[0]
            grads, touched = mean_grads[begin:begin + n], (accum_weights[begin:begin + n] > 0).reshape(-1)
            begin += n
            if iteration < opt.densify_until_iter:
                g.add_densification_stats(grads, touched)
                if iteration > opt.densify_from_iter and iteration % opt.densification_interval == 0:
                    info = g.densify_and_prune(opt, 20 if iteration > opt.opacity_reset_interval else None)
                    tot = [a + b for a, b in zip(tot, info)]
                if iteration % opt.opacity_reset_interval == 0:
                    g.reset_opacity()
            if iteration < opt.iterations:
                g.optimizer.step()
                g.optimizer.zero_grad(set_to_none=True)
        return tuple(tot)


class RangeFrames:
    """Range-image store with the accessors the loop uses (lib/scene/lidar_sensor.py: get_range_rays :395-434,
    get_depth / get_intensity / get_mask, inverse_projection_with_range :170-191); rays are given, not derived."""

    def __init__(self):
        self.rays: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
        self.depth: Dict[int, torch.Tensor] = {}
        self.intensity: Dict[int, torch.Tensor] = {}
        self.mask: Dict[int, torch.Tensor] = {}
        self.mask_index: Dict[int, torch.Tensor] = {}
        self.sensor_center: Dict[int, torch.Tensor] = {}

    @staticmethod
    def range_rays(H: int, W: int, inclination, sensor2world: torch.Tensor, data_type: str = "KITTI",
                   sensor2ego: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """The (H,W,3) ray grid of a spinning LiDAR range image (lib/scene/lidar_sensor.py:395-434): column w has azimuth
        (W-w-off)/W 2pi - pi - yaw, row h the inclination interpolated between the two bounds ((H-h-off)/H) or taken from a
        per-beam table (flipped: row 0 = last entry); ``off`` = 0.5 pixel and ``yaw`` = atan2 of the sensor-to-ego rotation
        for Waymo data (:42-49), 0 for KITTI; directions are rotated by ``sensor2world`` and normalised, the origin is
        its translation.  Float32 arithmetic in the reference's order (the golden ray fixtures are matched to 1e-6)."""
        dev = sensor2world.device
        f32 = dict(device=dev, dtype=torch.float32)
        waymo = data_type == "Waymo"
        off = 0.5 if waymo else 0.0
        yaw = torch.atan2(sensor2ego[1, 0], sensor2ego[0, 0]) if waymo else 0.0
        x = (torch.arange(W, 0, -1, **f32) - off) / float(W)
        gy, gx = torch.meshgrid(torch.ones(H, **f32), x, indexing="ij")
        azimuth = gx * 2 * torch.pi - torch.pi - yaw
        inc = list(inclination) if isinstance(inclination, (list, tuple)) else [-float(inclination), float(inclination)]
        if len(inc) == 2:
            gy = gy * (torch.arange(H, 0, -1, **f32).unsqueeze(-1) - off) / float(H)
            incl = gy * (inc[1] - inc[0]) + inc[0]
        else:
            incl = gy * torch.tensor(inc, **f32).flip(0).unsqueeze(-1)
        d = torch.stack([torch.cos(incl) * torch.cos(azimuth), torch.cos(incl) * torch.sin(azimuth), torch.sin(incl)], dim=-1)
        d = d @ sensor2world[:3, :3].T
        d = d / torch.norm(d, dim=-1, keepdim=True)
        o = sensor2world[:3, 3][None, None, :].expand(H, W, 3)
        return o.contiguous(), d.contiguous()

    def add_range_image(self, frame, depth, intensity, mask, inclination, sensor2world, data_type="KITTI", sensor2ego=None):
        """A frame given as range image + sensor pose (what the reference's loaders put into LiDARSensor)."""
        o, d = self.range_rays(depth.shape[0], depth.shape[1], inclination, sensor2world, data_type, sensor2ego)
        self.add_frame(frame, o, d, depth, intensity, mask)

    def add_frame(self, frame, rays_o, rays_d, depth, intensity, mask):
        self.rays[frame] = (rays_o, rays_d)
        self.depth[frame], self.intensity[frame], self.mask[frame] = depth, intensity, mask.bool()
        self.mask_index[frame] = torch.nonzero(mask.bool().reshape(-1)).squeeze(1)      # once per frame, not per iteration
        self.sensor_center[frame] = rays_o.reshape(-1, 3)[0]

    train_frames = property(lambda s: sorted(s.rays))
    get_range_rays = lambda s, f: s.rays[f]
    get_depth = lambda s, f: s.depth[f]
    get_intensity = lambda s, f: s.intensity[f]
    get_mask = lambda s, f: s.mask[f]

    def inverse_projection_with_range(self, frame, range_map, mask=None):
        """World points of the masked pixels; ``mask=None`` uses the frame's own mask through its cached index list
        (no boolean indexing, hence no device->host synchronisation inside the iteration)."""
        o, d = self.rays[frame]
        pts = (o + d * range_map.reshape(*d.shape[:2], 1)).reshape(-1, 3)
        if mask is None:
            return pts.index_select(0, self.mask_index[frame])
        return pts[mask.reshape(-1).bool()]


_BLUR_CACHE: Dict[tuple, torch.Tensor] = {}


def _blur_matrix(n: int, window_size: int, device, dtype) -> torch.Tensor:
    """(n, n) banded matrix of the normalised 1-D Gaussian window (sigma 1.5) with zero padding: ``x @ B`` blurs the last axis."""
    key = (n, window_size, str(device), dtype)
    m = _BLUR_CACHE.get(key)
    if m is None:
        g = torch.exp(-(torch.arange(window_size, dtype=torch.float32, device=device) - window_size // 2) ** 2 / (2 * 1.5 ** 2))
        g = (g / g.sum()).to(dtype)
        i = torch.arange(n, device=device)
        k = i[None, :] - i[:, None] + window_size // 2
        ok = (k >= 0) & (k < window_size)
        m = torch.zeros(n, n, device=device, dtype=dtype)
        m[ok] = g[k[ok]]
        if len(_BLUR_CACHE) > 16:
            _BLUR_CACHE.clear()
        _BLUR_CACHE[key] = m
    return m


def ssim(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11) -> torch.Tensor:
    """Mean structural similarity of (C,H,W) images, 11x11 Gaussian window with sigma 1.5 (loss_utils.py:45-89).  The reference
    convolves with the outer product of a 1-D window (zero padding); here the same separable blur is two small matrix products
    per image (banded matrices, rocBLAS) -- identical up to rounding, and the training loop does not depend on MIOpen's
    convolution search (one of its backward-data solvers faulted on the 16 x 256 test images)."""
    bh = _blur_matrix(img1.shape[-2], window_size, img1.device, img1.dtype)
    bw = _blur_matrix(img1.shape[-1], window_size, img1.device, img1.dtype)
    conv = lambda x: bh @ (x @ bw)
    mu1, mu2 = conv(img1), conv(img2)
    s1, s2, s12 = conv(img1 * img1) - mu1 * mu1, conv(img2 * img2) - mu2 * mu2, conv(img1 * img2) - mu1 * mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s1 + s2 + c2))).mean()


def training_step(scene: GaussianScene, frames: RangeFrames, frame, iteration: int, opt, background: torch.Tensor,
                  dynamic: bool = False, chamfer_points_detached: bool = True) -> Dict[str, torch.Tensor]:
    """One iteration of train.py:125-220: render, depth L1 + intensity L1/L2/DSSIM + ray-drop BCE + Chamfer + box
    regularisation, backward, then ``scene.optimize`` with ``means3D.grad`` and the accumulated hit weights.
    ``chamfer_points_detached``: the reference builds both point clouds from numpy (lidar_sensor.py:182-183), so its
    Chamfer term carries no gradient; False keeps the predicted points differentiable."""
    from .renderer import raytracing
    if opt.lambda_cd != 0:
        from .chamfer3D import chamfer_3DDist
    scene.update_learning_rate(iteration)
    if iteration % 1000 == 0:
        scene.oneupSHdegree()
    args = SimpleNamespace(dynamic=dynamic, opt=opt, pipe=SimpleNamespace())
    from . import renderer as _rnd

    def attempt():
        """render -> losses -> backward (train.py:148-214); everything the optimizer step reads afterwards"""
        pkg = raytracing(frame, scene.gaussians_assets, frames, background, args)
        depth, intensity, raydrop = pkg["depth"].squeeze(-1), pkg["intensity"].squeeze(-1), pkg["raydrop"]
        mask = frames.get_mask(frame)
        gt_depth, gt_int = frames.get_depth(frame), frames.get_intensity(frame)
        # masked means written as weighted sums: x[mask].mean() == (x * mask).sum() / mask.sum(), without the nonzero() + gather
        # (and its host synchronisation) that boolean indexing costs on every call
        mf = mask.to(depth.dtype)
        n_valid = mf.sum().clamp_min(1.0)
        mmean = lambda x: (x * mf).sum() / n_valid
        loss_depth = opt.lambda_depth_l1 * mmean(torch.abs(depth - gt_depth))
        loss_int = (opt.lambda_intensity_l1 * mmean(torch.abs(intensity - gt_int))
                    + opt.lambda_intensity_l2 * mmean((intensity - gt_int) ** 2)
                    + opt.lambda_intensity_dssim * (1 - ssim((intensity * mf).unsqueeze(0), (gt_int * mf).unsqueeze(0))))
        labels = (1.0 - mf).reshape(-1, 1)                                # 1 = dropped ray (train.py:188-193)
        loss_drop = opt.lambda_raydrop_bce * F.binary_cross_entropy(raydrop.reshape(-1, 1).clamp(1e-7, 1 - 1e-7), labels)
        if opt.lambda_cd != 0:
            pred_depth = depth.detach() if chamfer_points_detached else depth
            gt_pts = frames.inverse_projection_with_range(frame, gt_depth)
            pred_pts = frames.inverse_projection_with_range(frame, pred_depth)
            d1, d2, _, _ = chamfer_3DDist()(pred_pts[None].contiguous(), gt_pts[None].contiguous())
            loss_cd = opt.lambda_cd * (d1 + d2).mean() * 0.5
        else:                                                             # weight 0: the term (and its HIP operator) is skipped
            loss_cd = torch.zeros((), device=depth.device)
        loss_reg = sum(opt.lambda_reg * g.box_reg_loss() for g in scene.gaussians_assets)
        loss = loss_depth + loss_int + loss_drop + loss_cd + loss_reg
        loss.backward()
        return pkg, loss, loss_depth, loss_int, loss_drop, loss_cd

    pkg, loss, loss_depth, loss_int, loss_drop, loss_cd = attempt()
    redone = 0
    if _rnd.sharded is not None and getattr(opt, "verify_sharded_step", True):
        # Multi-GPU: a rank's culled build and the gradient exchange are sized SPECULATIVELY (no host wait inside the step).  A size that was too
        # small is flagged on the device -- but an optimizer step taken on that step's incomplete gradients could not be undone.  So the step is
        # verified HERE, between loss.backward() and scene.optimize (train.py:215-220): one wait for the status words every rank gathered.  A
        # reported problem re-runs the step once with exact sizes (the gradients of the failed attempt are dropped); a second one raises with
        # the parameters untouched.  Every rank takes the same branch: all ranks saw the same words.
        bad = _rnd.sharded.verify_step()
        if bad is not None:
            for g_ in scene.gaussians_assets:
                for p_ in g_._params().values():
                    p_.grad = None
            pkg, loss, loss_depth, loss_int, loss_drop, loss_cd = attempt()
            redone = 1
            bad = _rnd.sharded.verify_step()
            if bad is not None:
                from ._capi import LrtError
                for g_ in scene.gaussians_assets:                          #  the incomplete gradients of attempt 2 must not wait in .grad for a
                    for p_ in g_._params().values():                       # caller that catches this and runs the next backward on top of them
                        p_.grad = None
                raise LrtError(f"sharded training step {iteration}: incomplete results on a rank in two attempts ({bad[0]}; per-rank words {bad[1]}); "
                               "the optimizer step was NOT taken, the parameters are those of the previous iteration.  " + _rnd.sharded._STATUS_HELP)
    with torch.no_grad():
        info = scene.optimize(opt, iteration, pkg["means3D"].grad, pkg["accum_gaussian_weight"])
        # multi-GPU: the replicas are never synchronised (identical gradients + identical seeds keep them identical); verify it now and then
        k_chk = int(getattr(opt, "replica_check_interval", 500))
        if _rnd.sharded is not None and k_chk > 0 and iteration % k_chk == 0:
            _rnd.sharded.check_replicas([p_ for g in scene.gaussians_assets for p_ in g._params().values()])
    return {"loss": loss.detach(), "depth": loss_depth.detach(), "intensity": loss_int.detach(), "raydrop": loss_drop.detach(),
            "chamfer": loss_cd.detach(), "densify": info, "points": sum(g._xyz.shape[0] for g in scene.gaussians_assets), "step_redone": redone}
