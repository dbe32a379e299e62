#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run ONCE in the build
container, where /root/reference exists; the GPU box never runs this).

TEST INFRASTRUCTURE.  Two kinds of vectors are written:

1. *Reference pins* -- outputs of the reference's own importable Python
   helpers, executed on CPU through a shim (absent third-party modules are
   stubbed, ``device="cuda"`` is rewritten to ``"cpu"``):
     lib/utils/general_utils.py : build_rotation, quaternion_raw_multiply
     lib/utils/sh_utils.py      : eval_sh, RGB2SH
     lib/utils/primitive_utils.py: build2DRectangle
     lib/scene/lidar_sensor.py  : LiDARSensor.get_range_rays (KITTI + Waymo mode)
   -> tests/golden/conventions.npz, rays_*.npz
   lib/scene/gaussian_model.py : GaussianModel (training_setup, Adam groups, learning-rate schedule, add_densification_stats,
                                 densify_and_prune incl. the tracking-box rule, reset_opacity, capture) driven through the
                                 sequence of calls of SceneLidar.optimize (lib/dataloader/gs_loader.py:243-298)
   lib/utils/loss_utils.py    : l1_loss, l2_loss, ssim, BinaryCrossEntropyLoss
   eval.py                    : LiDARRTMeter.compute_raydrop_metrics / compute_fscore (the metric methods that need no
                                 third-party package; depth / intensity SSIM is skimage's, LPIPS a pretrained network)
   -> tests/golden/loop_golden.npz   (`--only-loop`)
2. *Oracle regression vectors* -- outputs of our CPU restatement
   (oracle/lrt_oracle.c, float32) on the seeded S10k scene
   -> tests/golden/s10k_golden.npz, s1m_stats.json.
   These are NOT outputs of the reference CUDA/OptiX kernels (which cannot
   run here): parity with the kernels themselves stays "unpinned".

Only data (inputs + expected outputs) is stored; no reference source text.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)


def _import_reference():
    import torch
    from torch.overrides import TorchFunctionMode

    class _Anything(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return _Anything(self.__name__ + "." + name)

        def __call__(self, *a, **k):
            return _Anything(self.__name__ + "()")

    for name in ["simple_knn", "simple_knn._C", "plyfile", "open3d", "icosphere", "cv2",
                 "tensorboardX", "lpips", "tqdm", "imageio", "matplotlib", "matplotlib.pyplot",
                 "matplotlib.cm", "trimesh", "skimage", "skimage.metrics", "roma", "kornia",
                 "PIL", "PIL.Image", "termcolor", "ruamel", "ruamel.yaml"]:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = _Anything(name)
    # the reference's console_utils pip-installs packages at import: stub it
    cu = _Anything("lib.utils.console_utils")
    sys.modules["lib.utils.console_utils"] = cu

    class CudaToCpu(TorchFunctionMode):
        def __torch_function__(self, func, types_, args=(), kwargs=None):
            kwargs = dict(kwargs or {})
            dev = kwargs.get("device", None)
            if dev is not None and "cuda" in str(dev):
                kwargs["device"] = "cpu"
            name = getattr(func, "__name__", "")
            if name == "cuda":
                return args[0]
            return func(*args, **kwargs)

    sys.path.insert(0, REF)
    mode = CudaToCpu()
    mode.__enter__()
    from lib.utils import general_utils, sh_utils  # noqa
    from lib.utils import primitive_utils  # noqa
    from lib.scene import lidar_sensor  # noqa
    return torch, general_utils, sh_utils, primitive_utils, lidar_sensor, mode


def gen_reference_pins():
    torch, gu, shu, pu, ls, mode = _import_reference()
    rng = np.random.default_rng(123)
    N = 128
    q = rng.normal(size=(N, 4)).astype(np.float32)          # un-normalised on purpose
    qt = torch.from_numpy(q)
    R = gu.build_rotation(qt).numpy()
    qa = rng.normal(size=(N, 4)).astype(np.float32)
    qprod = gu.quaternion_raw_multiply(None, torch.from_numpy(qa), qt).numpy()

    means = rng.normal(size=(N, 3)).astype(np.float32) * 10
    scales = np.exp(rng.uniform(np.log(0.03), np.log(0.25), (N, 2))).astype(np.float32)
    qn = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    opac = np.clip(1 / (1 + np.exp(-rng.normal(0, 2, (N, 1)))), 0.01, 0.99).astype(np.float32)
    verts, faces, _ = pu.build2DRectangle(torch.from_numpy(means), torch.from_numpy(scales),
                                          torch.from_numpy(qn), torch.from_numpy(opac))
    verts = verts.numpy(); faces = faces.numpy()

    dirs = rng.normal(size=(N, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dirs = dirs.astype(np.float64)
    sh = rng.normal(size=(N, 3, 16)).astype(np.float64)      # eval_sh layout (..., C, 16)
    sh_out = {}
    for deg in range(4):
        sh_out[f"eval_sh_deg{deg}"] = shu.eval_sh(deg, torch.from_numpy(sh), torch.from_numpy(dirs)).numpy()
    rgb = rng.uniform(0, 1, (N, 3))
    rgb2sh = shu.RGB2SH(torch.from_numpy(rgb)).numpy()

    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "conventions.npz"),
                        quats=q, build_rotation=R, quat_a=qa, quat_raw_multiply=qprod,
                        means=means, scales=scales, quats_unit=qn, opacities=opac,
                        rect_vertices=verts, rect_faces=faces,
                        sh_dirs=dirs, sh_coeffs_c16=sh, rgb=rgb, rgb2sh=rgb2sh, **sh_out)

    # ---- ray grids from LiDARSensor.get_range_rays
    def rays(H, W, data_type, bounds, s2w=None, s2e=None):
        s2e = np.eye(4, dtype=np.float32) if s2e is None else s2e
        sensor = ls.LiDARSensor(s2e, "top", bounds, data_type)
        sensor.H, sensor.W = H, W
        s2w = torch.eye(4) if s2w is None else torch.from_numpy(s2w)
        sensor.sensor2world[0] = s2w.float()
        sensor.sensor_center[0] = s2w[:3, 3].float()
        o, d = sensor.get_range_rays(0)
        assert not o.is_contiguous() or H * W == 1       # expanded view in the reference
        return o.contiguous().numpy(), d.contiguous().numpy()

    import math
    kb = [math.radians(-24.9), math.radians(2.0)]          # kitti_loader/__init__.py:187
    o, d = rays(16, 256, "KITTI", kb)
    np.savez_compressed(os.path.join(OUT, "rays_kitti_16x256.npz"), ray_o=o, ray_d=d,
                        inc_bounds_deg=np.array([-24.9, 2.0]))
    o, d = rays(64, 2048, "KITTI", kb)
    np.savez_compressed(os.path.join(OUT, "rays_kitti_64x2048_sample.npz"),
                        rows=np.arange(0, 64, 7), cols=np.arange(0, 2048, 61),
                        ray_d=d[::7, ::61], sha256=hashlib.sha256(d.tobytes()).hexdigest(),
                        inc_bounds_deg=np.array([-24.9, 2.0]))
    # a posed KITTI sensor and a Waymo-mode grid (per-beam inclination table + yaw offset)
    ang = 0.3
    s2w = np.eye(4, dtype=np.float32)
    s2w[:3, :3] = np.array([[math.cos(ang), -math.sin(ang), 0], [math.sin(ang), math.cos(ang), 0], [0, 0, 1]], np.float32)
    s2w[:3, 3] = [1.5, -2.0, 1.8]
    o, d = rays(8, 32, "KITTI", kb, s2w=s2w)
    np.savez_compressed(os.path.join(OUT, "rays_kitti_posed_8x32.npz"), ray_o=o, ray_d=d, sensor2world=s2w)
    beams = np.linspace(-0.31, 0.04, 8).astype(np.float32).tolist()
    o, d = rays(8, 40, "Waymo", beams, s2w=s2w, s2e=s2w)
    np.savez_compressed(os.path.join(OUT, "rays_waymo_8x40.npz"), ray_o=o, ray_d=d, sensor2world=s2w,
                        beam_inclinations=np.array(beams, np.float32))
    mode.__exit__(None, None, None)
    print("reference pins written to", OUT)


def gen_oracle_vectors(with_s1m: bool):
    from lidar_rt_amd import scenes
    from oracle import oracle
    sc, o, d = scenes.s10k()
    orc = oracle.Oracle(sc["means"], sc["scales"], sc["rotations"], sc["opacities"], "f32")
    fw = orc.forward(o, d, sc["shs"], 3, scenes.BG_DEFAULT, stats=True)
    dL = scenes.upstream_grad(16, 256)
    bw = orc.backward(o, d, sc["shs"], 3, scenes.BG_DEFAULT, fw["out"], dL)
    np.savez_compressed(os.path.join(OUT, "s10k_golden.npz"),
                        out=fw["out"], accum=fw["accum"], n_cand=fw["n_cand"], n_comp=fw["n_comp"],
                        d_means=bw["means"], d_shs=bw["shs"], d_opacities=bw["opacities"],
                        d_scales=bw["scales"], d_rotations=bw["rotations"],
                        scene_sha256=hashlib.sha256(b"".join(sc[k].tobytes() for k in sorted(sc))).hexdigest())
    print("s10k: C=%.3f K=%.3f" % (fw["n_cand"].mean(), fw["n_comp"].mean()))
    if with_s1m:
        sc, o, d = scenes.s1m()
        orc = oracle.Oracle(sc["means"], sc["scales"], sc["rotations"], sc["opacities"], "f32")
        fw = orc.forward(o, d, sc["shs"], 3, scenes.BG_DEFAULT, stats=True)
        nc, nk = fw["n_cand"], fw["n_comp"]
        C, K = float(nc.mean()), float(nk.mean())
        stats = {
            "scene": "S1M", "P": 1_000_000, "H": 64, "W": 2048, "sh_degree": 3, "seed": scenes.SEED,
            "C_mean_candidates_per_ray": C, "K_mean_composited_per_ray": K,
            "K_median": float(np.median(nk)), "K_p95": float(np.percentile(nk, 95)), "K_max": int(nk.max()),
            "C_max": int(nc.max()),
            "B_ray_bytes_fwd_bwd_deg3": 156 + 80 * C + 856 * K,
            "B_ray_bytes_fwd_deg3": 60 + 40 * C + 200 * K,
            "B_ray_bytes_bwd_deg3": 96 + 40 * C + 656 * K,
            "frac_saturated_T_lt_1e-3": float((fw["out"][..., 8] < 1e-3).mean()),
            "out_channel_means": [float(x) for x in fw["out"].reshape(-1, 9).mean(0)],
            "out_sha256_f32_oracle": hashlib.sha256(fw["out"].tobytes()).hexdigest(),
            "K_hist_bins_of_8": np.bincount(np.minimum(nk.ravel() // 8, 31), minlength=32).tolist(),
            "scene_sha256": hashlib.sha256(b"".join(sc[k].tobytes() for k in sorted(sc))).hexdigest(),
        }
        with open(os.path.join(OUT, "s1m_stats.json"), "w") as f:
            json.dump(stats, f, indent=1)
        print("s1m: C=%.3f K=%.3f" % (C, K))


def gen_preprocess_pins():
    """tests/golden/preprocess_golden.npz: the parameter pre-processing chain of the reference (activations, actor
    rigid transform, quaternion composition, concatenation over assets) evaluated with the reference's OWN functions
    (lib/utils/general_utils.py: build_rotation, quaternion_raw_multiply; the getters of lib/scene/gaussian_model.py:112-148
    are one-line compositions of torch.exp / torch.sigmoid / F.normalize and are called exactly like there), and torch
    autograd through them for a random upstream gradient."""
    torch, gu, shu, pu, ls, mode = _import_reference()
    import torch.nn.functional as F
    rng = np.random.default_rng(777)
    counts = [37, 11, 5, 20]                                  # background + 3 actors
    P = sum(counts)
    raw = {"xyz": rng.normal(size=(P, 3)).astype(np.float32) * 5, "log_scales": rng.normal(-2, 0.7, (P, 2)).astype(np.float32),
           "rot_raw": rng.normal(size=(P, 4)).astype(np.float32) * rng.uniform(0.2, 3, (P, 1)).astype(np.float32),
           "opacity_logit": rng.normal(0, 2, (P, 1)).astype(np.float32)}
    poses = np.zeros((len(counts), 8), np.float32); poses[:, 3] = 1
    for a in range(1, len(counts)):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        poses[a] = np.concatenate([rng.normal(size=3) * 10, q, [1.0]])
    seg = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    t = {k: torch.from_numpy(v).requires_grad_(True) for k, v in raw.items()}
    means, obj_rot, rot_local = [], [], []
    for a in range(len(counts)):
        s, e = int(seg[a]), int(seg[a + 1])
        xyz = t["xyz"][s:e]
        if poses[a, 7] != 0:                                  # get_world_xyz (gaussian_model.py:129-134)
            qa = torch.from_numpy(poses[a, 3:7]).reshape(1, 4)
            R = gu.build_rotation(qa).squeeze(0)
            means.append(xyz @ R.T + torch.from_numpy(poses[a, 0:3]))
        else:
            qa = torch.zeros((1, 4))
            means.append(xyz)
        obj_rot.append(qa.expand(e - s, -1))                  # gaussian_renderer/__init__.py:90-92
        rot_local.append(F.normalize(t["rot_raw"][s:e]))      # rotation_activation (gaussian_model.py:32,127)
    means3D = torch.cat(means, 0)
    opacity = torch.sigmoid(t["opacity_logit"]); scales = torch.exp(t["log_scales"])
    rots_bkgd = rot_local[0]                                  # gaussian_renderer/__init__.py:124-130 (dynamic, no decomp)
    rl = F.normalize(torch.cat(rot_local[1:], 0), dim=1)
    rotations = torch.cat([rots_bkgd, gu.quaternion_raw_multiply(None, torch.cat(obj_rot[1:], 0), rl)], 0)
    up = {k: rng.normal(size=v.shape).astype(np.float32) for k, v in
          (("means", means3D), ("scales", scales), ("rotations", rotations), ("opacities", opacity))}
    loss = (means3D * torch.from_numpy(up["means"])).sum() + (scales * torch.from_numpy(up["scales"])).sum() + \
           (rotations * torch.from_numpy(up["rotations"])).sum() + (opacity * torch.from_numpy(up["opacities"])).sum()
    loss.backward()
    np.savez_compressed(os.path.join(OUT, "preprocess_golden.npz"), seg_start=seg, poses=poses,
                        **{"in_" + k: v for k, v in raw.items()},
                        out_means=means3D.detach().numpy(), out_scales=scales.detach().numpy(),
                        out_rotations=rotations.detach().numpy(), out_opacities=opacity.detach().numpy(),
                        **{"up_" + k: v for k, v in up.items()},
                        **{"grad_" + k: t[k].grad.numpy() for k in raw})
    print("preprocess_golden.npz written")


def gen_loop_pins():
    """tests/golden/loop_golden.npz: the training-loop bookkeeping of the reference executed with the reference's own classes on CPU
    (inputs + expected outputs only).  lidar_rt_amd/training.py and evaluation.py are tested against it (tests/test_loop_golden.py)."""
    torch, gu, shu, pu, ls, mode = _import_reference()
    from types import SimpleNamespace
    from torch import nn
    from lib.scene import gaussian_model as gm
    from lib.utils import loss_utils as lu
    from lidar_rt_amd.training import default_options        # the hyper-parameter VALUES (configs/base.yaml, configs/exp.yaml) as one namespace
    out = {}
    opt = default_options()
    opt.densify_grad_threshold, opt.densify_scale_threshold, opt.thresh_opa_prune = 4e-4, 0.012, 0.2
    out["opt_densify_grad_threshold"], out["opt_densify_scale_threshold"], out["opt_thresh_opa_prune"] = 4e-4, 0.012, 0.2

    def make(P, seed, bbox=None):
        g = torch.Generator().manual_seed(seed)
        r = lambda *s: torch.randn(*s, generator=g)
        m = gm.GaussianModel(2, 3, extent=10.0, bounding_box=bbox)
        m.spatial_lr_scale = 10.0                               # create_from_pcd sets it to the extent (gaussian_model.py:156)
        init = {"xyz": r(P, 3) * 2.0, "f_dc": r(P, 1, 3), "f_rest": r(P, 15, 3) * 0.1, "opacity": r(P, 1), "scaling": r(P, 2) * 0.7 - 2.2, "rotation": r(P, 4)}
        m._xyz, m._features_dc, m._features_rest = nn.Parameter(init["xyz"].clone()), nn.Parameter(init["f_dc"].clone()), nn.Parameter(init["f_rest"].clone())
        m._opacity, m._scaling, m._rotation = nn.Parameter(init["opacity"].clone()), nn.Parameter(init["scaling"].clone()), nn.Parameter(init["rotation"].clone())
        m.max_radii2D = torch.zeros(P)
        return m, init, g

    def params(m):
        return {"xyz": m._xyz, "f_dc": m._features_dc, "f_rest": m._features_rest, "opacity": m._opacity, "scaling": m._scaling, "rotation": m._rotation}

    def drive(tag, m, g, iters, densify_at, reset_at, size_threshold):
        """The call sequence of SceneLidar.optimize for one asset, with seeded stand-ins for loss.backward()'s gradients."""
        m.training_setup(opt)
        out[tag + "_group_names"] = np.array([gr["name"] for gr in m.optimizer.param_groups])
        out[tag + "_group_lrs"] = np.array([gr["lr"] for gr in m.optimizer.param_groups], np.float64)
        out[tag + "_adam_eps"] = np.float64(m.optimizer.defaults["eps"])
        its = [0, 1, 10, 1000, 15000, 30000, 40000]
        out[tag + "_lr_iters"] = np.array(its); out[tag + "_lr_xyz"] = np.array([m.update_learning_rate(i) for i in its], np.float64)
        log = []
        for it in range(1, iters + 1):
            m.update_learning_rate(it)
            P = m._xyz.shape[0]
            grads = {n: torch.randn(p_.shape, generator=g) * 0.01 for n, p_ in params(m).items()}
            mean_grads = torch.randn(P, 3, generator=g) * 1e-3
            accum = (torch.rand(P, 1, generator=g) < 0.6).float() * torch.rand(P, 1, generator=g)
            for n, p_ in params(m).items():
                p_.grad = grads[n].clone()
                out[f"{tag}_it{it}_grad_{n}"] = grads[n].numpy()
            out[f"{tag}_it{it}_mean_grads"] = mean_grads.numpy(); out[f"{tag}_it{it}_accum"] = accum.numpy()
            with torch.no_grad():
                m.add_densification_stats(mean_grads, accum > 0)
                info = (0, 0, 0, 0)
                if it == densify_at:
                    torch.manual_seed(4242)                          # the split's (and the box rule's) torch.normal draws
                    info = m.densify_and_prune(opt, 0.005, size_threshold)
                if it == reset_at:
                    m.reset_opacity()
                m.optimizer.step()
                m.optimizer.zero_grad(set_to_none=True)
            log.append([m._xyz.shape[0]] + [int(x) for x in info])
            for n, p_ in params(m).items():
                out[f"{tag}_it{it}_param_{n}"] = p_.detach().numpy().copy()
            st = m.optimizer.state
            for n, p_ in params(m).items():
                if p_ in st:
                    out[f"{tag}_it{it}_m_{n}"] = st[p_]["exp_avg"].numpy().copy(); out[f"{tag}_it{it}_v_{n}"] = st[p_]["exp_avg_sq"].numpy().copy()
            out[f"{tag}_it{it}_grad_accum"] = m.xyz_gradient_accum.numpy().copy(); out[f"{tag}_it{it}_denom"] = m.denom.numpy().copy()
        out[tag + "_log"] = np.array(log)
        cap = m.capture()
        out[tag + "_capture_types"] = np.array([type(c).__name__ for c in cap])
        out[tag + "_capture_shapes"] = np.array([str(tuple(c.shape)) if hasattr(c, "shape") else "" for c in cap])
        out[tag + "_capture_state_keys"] = np.array(sorted(cap[10].keys()))
        return m

    # A: the background asset (no tracking box), densification at iteration 3 with the size rule, opacity reset at 5
    m, init, g = make(200, 11)
    for n, v in init.items(): out["A_init_" + n] = v.numpy()
    drive("A", m, g, 6, 3, 5, 20)
    # B: an actor with a tracking box (points outside the box are pruned with random samples), densification at 2
    from lib.scene.bounding_box import BoundingBox
    bb = BoundingBox("car", 7, np.array([3.0, 2.5, 2.0], np.float32))
    m, init, g = make(150, 12, bb)
    for n, v in init.items(): out["B_init_" + n] = v.numpy()
    out["B_box_size"] = np.array([3.0, 2.5, 2.0], np.float32)
    drive("B", m, g, 3, 2, -1, 20)
    with torch.no_grad():
        out["B_box_reg_loss"] = np.float64(m.box_reg_loss())

    # losses (lib/utils/loss_utils.py) on seeded images
    g = torch.Generator().manual_seed(5)
    a = torch.rand(1, 16, 64, generator=g); b = (a + 0.1 * torch.randn(1, 16, 64, generator=g)).clamp(0, 1)
    out["loss_img_a"], out["loss_img_b"] = a.numpy(), b.numpy()
    out["loss_l1"], out["loss_l2"], out["loss_ssim"] = np.float64(lu.l1_loss(a, b)), np.float64(lu.l2_loss(a, b)), np.float64(lu.ssim(a, b))
    labels = (torch.rand(1024, 1, generator=g) < 0.3); preds = torch.rand(1024, 1, generator=g).clamp(1e-4, 1 - 1e-4)
    out["bce_labels"], out["bce_preds"] = labels.numpy(), preds.numpy()
    out["loss_bce"] = np.float64(lu.BinaryCrossEntropyLoss()(labels, preds=preds))
    # eval.py's own metric methods (no third-party package inside)
    try:
        for name in ["diff_lidar_tracer", "tensorflow", "waymo_open_dataset", "lib.utils.chamfer3D", "lib.utils.chamfer3D.dist_chamfer_3D", "lib.gaussian_renderer",
                     "lib.dataloader", "lib.arguments", "lib.scene.unet", "lib.utils.image_utils"]:
            if name not in sys.modules:
                sys.modules[name] = type(sys.modules["plyfile"])(name) if "plyfile" in sys.modules and not hasattr(sys.modules["plyfile"], "__file__") else types.ModuleType(name)
        import importlib.util
        spec = importlib.util.spec_from_file_location("ref_eval", os.path.join(REF, "eval.py"))
        ev = importlib.util.module_from_spec(spec); spec.loader.exec_module(ev)
        me = SimpleNamespace(raydrop_ratio=0.4)
        rng = np.random.default_rng(9)
        gt = (rng.uniform(size=(16, 64, 1)) < 0.3).astype(np.float64); pr = np.where(rng.uniform(size=gt.shape) < 0.85, gt, 1 - gt)
        out["eval_drop_gt"], out["eval_drop_pred"] = gt, pr
        out["eval_raydrop_metrics"] = np.array(ev.LiDARRTMeter.compute_raydrop_metrics(me, gt, pr), np.float64)
        d1 = torch.from_numpy(rng.uniform(0, 0.2, (1, 500)) ** 2); d2 = torch.from_numpy(rng.uniform(0, 0.3, (1, 400)) ** 2)
        f, p1, p2 = ev.LiDARRTMeter.compute_fscore(me, d1, d2, threshold=0.05)
        out["eval_dist1"], out["eval_dist2"] = d1.numpy(), d2.numpy()
        out["eval_fscore"] = np.array([float(f[0]), float(p1[0]), float(p2[0])])
        print("eval.py metric methods pinned")
    except Exception as ex:                                        # eval.py drags half the repository in at import time
        print("eval.py not importable under the shim, its metric methods stay unpinned:", repr(ex)[:200])
    mode.__exit__(None, None, None)
    np.savez_compressed(os.path.join(OUT, "loop_golden.npz"), **out)
    print("loop_golden.npz written:", len(out), "arrays")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only-loop", action="store_true")
    ap.add_argument("--only-preprocess", action="store_true")
    ap.add_argument("--skip-reference", action="store_true")
    ap.add_argument("--s1m", action="store_true", help="also compute S1M statistics (needs ~1 min)")
    a = ap.parse_args()
    if a.only_loop:
        gen_loop_pins()
        sys.exit(0)
    if a.only_preprocess:
        gen_preprocess_pins()
        sys.exit(0)
    if not a.skip_reference:
        gen_reference_pins()
        gen_preprocess_pins()
        gen_loop_pins()
    gen_oracle_vectors(a.s1m)
