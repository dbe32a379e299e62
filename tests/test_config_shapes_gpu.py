"""GPU parity at the dataset shapes of BASELINE configs[2..4] (the data themselves are not available: synthetic frames with the
datasets' range-image geometry, ``lidar_rt_amd/scenes.py``):

  configs[2]  Waymo static   : ~2 M background Gaussians, 64 x 2650 top-LiDAR grid (per-beam inclinations, posed sensor)
                               (lib/dataloader/waymo_loader/__init__.py:36-131) + training_step iterations at that shape
  configs[3]  KITTI-360 dyn. : 66 x 1030 (lib/dataloader/kitti_loader/__init__.py:186), background + 8 rigid actor assets through
                               renderer.raytracing (lib/gaussian_renderer/__init__.py:76-160), poses changing per frame,
                               refit_interval > 0
  configs[4]  Waymo dyn. 4 M : ~4 M Gaussians under the Waymo grid (single GPU here; the 8-GPU split is the driver's)

Every comparison is HIP (through the drop-in API -> ctypes -> C ABI) against the fp32 oracle, bounded by k x the fp32-vs-fp64
oracle floor; the measured numbers go to gpurun_out/parity/*.json (committed digest: profiles/r02_parity.json).
"""
import math
import types

import numpy as np
import pytest
import torch

from lidar_rt_amd import scenes
from oracle import oracle

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from tests.hip_util import run_hip, parity_report, rel_l2, frac_outside
    from tests.test_renderer_gpu import Asset, DEV

GRADS = ("means", "scales", "rotations", "opacities", "shs")


def oracle_pair(sc, o, d, deg, bg, dL, prec):
    orc = oracle.Oracle(sc["means"], sc["scales"], sc["rotations"], sc["opacities"], prec)
    fw = orc.forward(o, d, sc["shs"], deg, bg, stats=True)
    bw = orc.backward(o, d, sc["shs"], deg, bg, fw["out"], dL)
    return fw, bw


def test_scene_module_ray_grids_match_the_training_loop_sensor():
    from lidar_rt_amd.training import RangeFrames
    H, W = scenes.WAYMO_HW
    s2w = scenes.pose_matrix((0.8, 0.1, 0.45), yaw=0.43, pitch=0.01, roll=-0.008); s2e = scenes.pose_matrix((1.43, 0, 2.18), yaw=0.02)
    o, d = scenes.range_rays(H, W, scenes.waymo_inclinations(H), s2w, "Waymo", s2e)
    o2, d2 = RangeFrames.range_rays(H, W, [float(x) for x in scenes.waymo_inclinations(H)], torch.as_tensor(s2w), "Waymo", torch.as_tensor(s2e))
    assert np.abs(d - d2.numpy()).max() < 1e-6 and np.array_equal(o, o2.numpy())
    H, W = scenes.KITTI360_HW
    o, d = scenes.range_rays(H, W, (math.radians(-24.9), math.radians(2.0)), np.eye(4, dtype=np.float32), "KITTI")
    assert np.array_equal(d, scenes.kitti_rays(H, W)[1])


def test_waymo_static_2m_matches_oracle_within_the_floor():
    """configs[2] shape: 2 M Gaussians, 64 x 2650 rays from a posed Waymo-style sensor."""
    sc, o, d = scenes.waymo_frame()
    H, W = o.shape[:2]
    assert (H, W) == (64, 2650) and sc["means"].shape[0] == 2_000_000
    dL = scenes.upstream_grad(H, W)
    f32 = oracle_pair(sc, o, d, 3, scenes.BG_DEFAULT, dL, "f32")
    f64 = oracle_pair(sc, o, d, 3, scenes.BG_DEFAULT, dL, "f64")
    h = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL)
    np.testing.assert_allclose(h["out"][..., 4] + h["out"][..., 8], 1.0, atol=2e-5)             # energy per ray
    from tests.event_gate import event_masked_gate
    event_masked_gate("waymo_static_2m", sc, o, d, 3, scenes.BG_DEFAULT, dL, f32[0], f64[0])
    parity_report("waymo_static_2m", h, f32, f64, f64_k=None, extra={
        "config": "BASELINE configs[2] shape: 2,000,000 Gaussians, 64x2650 Waymo-style grid", "rays": [H, W], "gaussians": 2_000_000,
        "C_mean": float(f32[0]["n_cand"].mean()), "K_mean": float(f32[0]["n_comp"].mean()), "K_max": int(f32[0]["n_comp"].max())})


def test_waymo_dynamic_4m_matches_oracle_within_the_floor():
    """configs[4] shape on one GPU: ~4 M Gaussians (background + posed actors), 64 x 2650 rays."""
    sc, o, d = scenes.waymo_dynamic_4m()
    H, W = o.shape[:2]
    P = sc["means"].shape[0]
    assert P == 4_000_000
    dL = scenes.upstream_grad(H, W)
    f32 = oracle_pair(sc, o, d, 3, scenes.BG_DEFAULT, dL, "f32")
    f64 = oracle_pair(sc, o, d, 3, scenes.BG_DEFAULT, dL, "f64")
    h = run_hip(sc, o, d, 3, scenes.BG_DEFAULT, dL)
    # fp64-arbitrated gate: asserted at the claim's own (1.1, 1.25) with the threshold events named, counted, certified and masked (tests/event_gate.py;
    # round 5 gated the L2 of this scene at a factor of 3.5 read off its own measurement); the unmasked table keeps the floor-relative bounds
    from tests.event_gate import event_masked_gate
    event_masked_gate("waymo_dynamic_4m", sc, o, d, 3, scenes.BG_DEFAULT, dL, f32[0], f64[0])
    parity_report("waymo_dynamic_4m", h, f32, f64, f64_k=None, extra={
        "config": "BASELINE configs[4] shape on one GPU: 4,000,000 Gaussians, 64x2650 Waymo-style grid", "rays": [H, W], "gaussians": P,
        "C_mean": float(f32[0]["n_cand"].mean()), "K_mean": float(f32[0]["n_comp"].mean()), "K_max": int(f32[0]["n_comp"].max())})


def _cpu_reference_frame(bg, actors, poses, o, d, w_depth, w_int, w_drop, prec="f32"):
    """The renderer's chain on the CPU: oracle/preprocess_ref (torch autograd) -> C oracle -> channel semantics of
    lib/gaussian_renderer/__init__.py:163-181 (use_rayhit off: raydrop = sigmoid(out[..., 2])).  Returns the rendered channels,
    the world-space means gradient and the raw-parameter gradients per asset."""
    from oracle import preprocess_ref
    parts = [bg] + list(actors)
    raw = {"xyz": [], "ls": [], "rot": [], "opl": []}
    for p in parts:
        raw["xyz"].append(torch.tensor(p["means"])); raw["ls"].append(torch.tensor(np.log(p["scales"])))
        raw["rot"].append(torch.tensor(p["rotations"] * 1.7))
        op = p["opacities"]; raw["opl"].append(torch.tensor(np.log(op / (1 - op))))
    cat = {k: torch.cat(v, 0).requires_grad_(True) for k, v in raw.items()}
    seg = np.cumsum([0] + [p["means"].shape[0] for p in parts])
    tab = torch.zeros((len(parts), 8))
    for a, (t, q) in enumerate(poses):
        tab[a + 1, 0:3] = torch.tensor(t); tab[a + 1, 3:7] = torch.tensor(q); tab[a + 1, 7] = 1.0
    means, sc_, rot, opac = preprocess_ref.preprocess(cat["xyz"], cat["ls"], cat["rot"], cat["opl"], seg, tab)
    shs = np.concatenate([p["shs"] for p in parts], 0)
    S = {"means": means.detach().numpy(), "scales": sc_.detach().numpy(), "rotations": rot.detach().numpy(),
         "opacities": opac.detach().numpy(), "shs": shs}
    orc = oracle.Oracle(S["means"], S["scales"], S["rotations"], S["opacities"], prec)
    fw = orc.forward(o, d, shs, 3, scenes.BG_DEFAULT)
    out = fw["out"].astype(np.float64)
    sig = 1.0 / (1.0 + np.exp(-out[..., 2]))
    dL = np.zeros_like(fw["out"])
    dL[..., 0] = w_int; dL[..., 3] = w_depth; dL[..., 2] = w_drop * sig * (1 - sig)
    bw = orc.backward(o, d, shs, 3, scenes.BG_DEFAULT, fw["out"], dL)
    gr = torch.autograd.grad([means, sc_, rot, opac], [cat["xyz"], cat["ls"], cat["rot"], cat["opl"]],
                             [torch.tensor(bw["means"].astype(np.float32)), torch.tensor(bw["scales"].astype(np.float32)),
                              torch.tensor(bw["rotations"].astype(np.float32)), torch.tensor(bw["opacities"].astype(np.float32))])
    raw_g = {"means": gr[0].numpy(), "scales": gr[1].numpy(), "rotations": gr[2].numpy(), "opacities": gr[3].numpy(), "shs": bw["shs"]}
    return {"out": fw["out"], "accum": fw["accum"], "raydrop": sig, "world_means_grad": bw["means"], "world_scene": S, "dL": dL}, raw_g


def test_kitti360_dynamic_actors_with_refit_match_oracle():
    """configs[3] shape: 66 x 1030, background + 8 actor assets through renderer.raytracing, poses and sensor moving per frame,
    bvh_refit_interval = 3 (frames 0 and 4 build, 1-3 refit).  Every frame is checked against the CPU chain; frame 2 (a refit
    frame) also against the fp64 floor."""
    from lidar_rt_amd import renderer
    bg, actors, poses_of, rays_of = scenes.kitti360_dynamic()
    P_bg = bg["means"].shape[0]
    assets = [Asset(bg, slice(None))] + [Asset(a, slice(None), pose=(np.zeros(3, np.float32), np.array([1, 0, 0, 0], np.float32))) for a in actors]
    args = types.SimpleNamespace(dynamic=True, opt=types.SimpleNamespace(use_rayhit=False, bvh_refit_interval=3), pipe=types.SimpleNamespace())
    H, W = scenes.KITTI360_HW
    rng = np.random.default_rng(77)
    w_depth = (rng.normal(size=(H, W)) / (H * W)).astype(np.float32); w_int = (rng.normal(size=(H, W)) / (H * W)).astype(np.float32)
    w_drop = (rng.normal(size=(H, W)) / (H * W)).astype(np.float32)
    renderer.tracer_2dgs = None                                   # a fresh tracer: the refit counter starts with this test
    built = []
    try:
        for frame in range(5):
            o, d = rays_of(frame)
            poses = poses_of(frame)
            for a, (t, q) in zip(assets[1:], poses):
                a.bounding_box.frame = {frame: (torch.as_tensor(t, device=DEV), torch.as_tensor(q, device=DEV).reshape(1, 4), None, None)}
            for a in assets:
                for p in a.params():
                    p.grad = None
            sensor = (torch.as_tensor(o, device=DEV), torch.as_tensor(d, device=DEV), torch.as_tensor(o[0, 0], device=DEV))
            res = renderer.raytracing(frame, assets, sensor, torch.tensor(scenes.BG_DEFAULT), args)
            loss = (res["depth"].squeeze(-1) * torch.as_tensor(w_depth, device=DEV)).sum() + \
                   (res["intensity"].squeeze(-1) * torch.as_tensor(w_int, device=DEV)).sum() + \
                   (res["raydrop"].squeeze(-1) * torch.as_tensor(w_drop, device=DEV)).sum()
            loss.backward()
            torch.cuda.synchronize()
            built.append(renderer.tracer_2dgs.optix_context._since_full.get(0))
            ref, raw_g = _cpu_reference_frame(bg, actors, poses, o, d, w_depth, w_int, w_drop)
            hip_out = np.zeros((H, W, 9), np.float32)
            hip_out[..., 0] = res["intensity"].squeeze(-1).detach().cpu().numpy(); hip_out[..., 3] = res["depth"].squeeze(-1).detach().cpu().numpy()
            got_g = {"means": np.concatenate([a._xyz.grad.cpu().numpy() for a in assets]), "scales": np.concatenate([a._scaling.grad.cpu().numpy() for a in assets]),
                     "rotations": np.concatenate([a._rotation.grad.cpu().numpy() for a in assets]),
                     "opacities": np.concatenate([a._opacity.grad.cpu().numpy() for a in assets]),
                     "shs": np.concatenate([a._features.grad.cpu().numpy() for a in assets])}
            # every frame against the CPU chain in float32, bounded by the float32-vs-float64 floor of the same chain
            ref64, raw64 = _cpu_reference_frame(bg, actors, poses, o, d, w_depth, w_int, w_drop, prec="f64")
            assert rel_l2(res["raydrop"].squeeze(-1).detach().cpu().numpy(), ref["raydrop"]) < max(4 * rel_l2(ref["raydrop"], ref64["raydrop"]), 2e-4), frame
            assert rel_l2(res["means3D"].grad.cpu().numpy(), ref["world_means_grad"]) < max(4 * rel_l2(ref["world_means_grad"], ref64["world_means_grad"]), 2e-3), frame
            hip = {"out": np.array(ref["out"]), "accum": res["accum_gaussian_weight"].squeeze(-1).detach().cpu().numpy(), "grads": got_g}
            hip["out"][..., 0] = hip_out[..., 0]; hip["out"][..., 3] = hip_out[..., 3]     # the channels the renderer exposes unchanged
            # the renderer's chain against the CPU chain: floor-relative bounds asserted, fp64-arbitrated ratios recorded; the claim itself -- (1.1, 1.25),
            # no scene factor (round 5 gated this scene's L2 at 3.5) -- is asserted on the posed world-space scene of a build frame and of a refit frame with
            # the threshold events named, counted, certified and masked (tests/event_gate.py)
            if frame in (0, 2):
                from tests.event_gate import event_masked_gate
                event_masked_gate(f"kitti360_dynamic_frame{frame}", ref["world_scene"], o, d, 3, scenes.BG_DEFAULT, ref["dL"], None, None)
            parity_report(f"kitti360_dynamic_frame{frame}_{'build' if built[-1] == 0 else 'refit'}", hip,
                          ({"out": ref["out"], "accum": ref["accum"]}, raw_g), ({"out": ref64["out"], "accum": ref64["accum"]}, raw64), f64_k=None,
                          extra={"config": "BASELINE configs[3] shape: 66x1030, 500k background + 8 actors x 8k through renderer.raytracing "
                                           "(bvh_refit_interval=3); gradients w.r.t. the RAW parameters of all assets", "rays": [H, W],
                                 "gaussians": P_bg + 8 * 8000, "note": "out.* rows other than intensity / depth are the oracle against itself"})
    finally:
        renderer.tracer_2dgs = None
    assert built == [0, 1, 2, 3, 0], built                        # build, three refits, build


def test_training_steps_at_the_waymo_static_shape():
    """configs[2]: the train.py-shaped loop (render, all losses, backward, Adam, densification bookkeeping) at the dataset's
    size: 2 M Gaussians, 64 x 2650 rays.  Ground truth = the scene's own rendering with perturbed parameters, so the loss has
    something to do; checks that it runs, stays finite and goes down."""
    from lidar_rt_amd import training
    from lidar_rt_amd.renderer import raytracing
    sc, o, d = scenes.waymo_frame(P=2_000_000)
    H, W = o.shape[:2]
    t = lambda a: torch.as_tensor(a, device=DEV)
    opt = training.default_options()
    asset = training.GaussianAsset.from_tensors(t(sc["means"]), t(sc["shs"][:, :1]).contiguous(), t(sc["shs"][:, 1:]).contiguous(),
                                                torch.log(t(sc["scales"])), t(sc["rotations"]),
                                                training.inverse_sigmoid(t(sc["opacities"])), max_sh_degree=3, extent=75.0)
    asset.active_sh_degree = 3
    scene = training.GaussianScene([asset])
    frames = training.RangeFrames()
    bgc = torch.tensor(scenes.BG_DEFAULT, device=DEV)
    args = types.SimpleNamespace(dynamic=False, opt=opt, pipe=types.SimpleNamespace())
    with torch.no_grad():
        for f in range(2):
            of, df = scenes.waymo_sensor_rays(f)
            pkg = raytracing(f, [asset], (t(of), t(df), t(of[0, 0])), bgc, args)
            depth = pkg["depth"].squeeze(-1); inten = pkg["intensity"].squeeze(-1)
            mask = pkg["raydrop"].squeeze(-1) < 0.5
            frames.add_frame(f, t(of), t(df), depth.clone(), inten.clone(), mask)
        asset._xyz.add_(0.01 * torch.randn_like(asset._xyz)); asset._opacity.add_(0.3 * torch.randn_like(asset._opacity))
    scene.training_setup(opt)
    losses = []
    for it in range(1, 9):
        info = training.training_step(scene, frames, (it - 1) % 2, it, opt, bgc)
        losses.append(float(info["loss"]))
    torch.cuda.synchronize()
    assert all(np.isfinite(losses)), losses
    assert min(losses[-2:]) < max(losses[:2]), losses
