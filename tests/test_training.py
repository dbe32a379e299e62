"""Training-loop counterpart (lidar_rt_amd/training.py): optimiser surgery, densification rules and the checkpoint layout
on CPU; a short optimisation run through the whole MI355X path on the GPU."""
import io
import types

import numpy as np
import pytest
import torch

from lidar_rt_amd import scenes, training


def make_asset(P=200, seed=0, device="cpu", **kw):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    a = training.GaussianAsset.from_tensors(r(P, 3) * 5, r(P, 1, 3), r(P, 15, 3) * 0.1, r(P, 2) * 0.3 - 2.0, r(P, 4), r(P, 1),
                                            extent=10.0, **kw)
    for n in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
        setattr(a, n, torch.nn.Parameter(getattr(a, n).detach().to(device).requires_grad_(True)))
    a.max_radii2D = a.max_radii2D.to(device)
    return a


def fake_step(a):
    for p in a._params().values():
        p.grad = torch.randn_like(p) * 0.01
    a.optimizer.step()


def test_adam_groups_and_learning_rate_schedule():
    opt = training.default_options()
    a = make_asset()
    a.training_setup(opt)
    names = [g["name"] for g in a.optimizer.param_groups]
    assert names == ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]           # gaussian_model.py:193-200
    lrs = {g["name"]: g["lr"] for g in a.optimizer.param_groups}
    assert lrs["f_rest"] == pytest.approx(opt.feature_lr / 20) and lrs["xyz"] == pytest.approx(opt.position_lr_init * 10.0)
    assert a.optimizer.defaults["eps"] == 1e-15
    assert a.update_learning_rate(0) == pytest.approx(opt.position_lr_init * 10.0)
    assert a.update_learning_rate(opt.position_lr_max_steps) == pytest.approx(opt.position_lr_final * 10.0)
    mid = a.update_learning_rate(opt.position_lr_max_steps // 2)
    assert mid == pytest.approx(10.0 * np.sqrt(opt.position_lr_init * opt.position_lr_final), rel=1e-6)   # log-linear


def test_prune_append_and_reset_keep_the_optimizer_state_consistent():
    opt = training.default_options()
    a = make_asset(300)
    a.training_setup(opt)
    fake_step(a)
    m0 = a.optimizer.state[a._xyz]["exp_avg"].clone()
    mask = torch.zeros(300, dtype=torch.bool); mask[::3] = True
    a.prune_points(mask)
    assert a._xyz.shape[0] == 200 and a.denom.shape == (200, 1) and a.max_radii2D.shape == (200,)
    for name, p in a._params().items():
        st = a.optimizer.state[p]
        assert st["exp_avg"].shape == p.shape and st["exp_avg_sq"].shape == p.shape, name
    assert torch.equal(a.optimizer.state[a._xyz]["exp_avg"], m0[~mask])
    a._append(a._select(torch.arange(200) < 10))
    assert a._xyz.shape[0] == 210 and torch.all(a.optimizer.state[a._xyz]["exp_avg"][200:] == 0)
    a.reset_opacity()
    assert float(a.get_opacity.detach().max()) <= 0.01 + 1e-6 and torch.all(a.optimizer.state[a._opacity]["exp_avg"] == 0)
    fake_step(a)                                                       # the rebuilt optimiser still steps


def test_densify_clone_split_prune_rules():
    opt = training.default_options()
    a = make_asset(400, seed=3)
    a.training_setup(opt)
    fake_step(a)
    with torch.no_grad():
        a._scaling[:200] = np.log(1e-4)                                # small -> clone candidates (threshold 2e-4 * extent 10 = 2e-3)
        a._scaling[200:] = np.log(0.05)                                # large -> split candidates
        a._opacity[:] = 2.0
        a._opacity[150] = -9.0                                         # transparent (and not densified) -> pruned
    a.xyz_gradient_accum[:] = 0; a.denom[:] = 1
    a.xyz_gradient_accum[0:50] = 1.0                                   # above densify_grad_threshold
    a.xyz_gradient_accum[200:230] = 1.0
    torch.manual_seed(0)
    before = a._xyz.detach().clone()
    n_clone, n_split, n_scale, n_opa = a.densify_and_prune(opt, None)
    assert (n_clone, n_split, n_scale, n_opa) == (50, 30, 0, 1)
    assert a._xyz.shape[0] == 400 + 50 + 2 * 30 - 30 - 1               # clones added, each split parent replaced by 2 children
    kept = torch.ones(400, dtype=torch.bool); kept[200:230] = False; kept[150] = False
    assert torch.equal(a._xyz.detach()[:int(kept.sum())], before[kept])          # survivors keep their order
    assert float(a.get_scaling.max()) <= 0.05 + 1e-6
    for name, p in a._params().items():
        assert a.optimizer.state[p]["exp_avg"].shape == p.shape, name
    assert a.denom.shape[0] == a._xyz.shape[0] and float(a.denom.sum()) == 0


def test_actor_box_pruning_and_regulariser():
    opt = training.default_options()
    box = types.SimpleNamespace(frame={}, min_xyz=torch.tensor([-1.0, -1, -1]), max_xyz=torch.tensor([1.0, 1, 1]))
    a = make_asset(100, seed=5, bounding_box=box)
    a.training_setup(opt)
    with torch.no_grad():
        a._xyz[:] = 0.0; a._xyz[:10] = 50.0                            # ten Gaussians far outside the tracking box
        a._scaling[:] = np.log(1e-3); a._opacity[:] = 2.0
    a.xyz_gradient_accum[:] = 0; a.denom[:] = 1
    assert float(a.box_reg_loss()) > 0
    a.densify_and_prune(opt, 20)
    assert a._xyz.shape[0] == 90


def test_checkpoint_tuple_layout_roundtrip():
    opt = training.default_options()
    sc = training.GaussianScene([make_asset(50, 1), make_asset(20, 2)])
    sc.training_setup(opt)
    for g in sc.gaussians_assets:
        fake_step(g)
    buf = io.BytesIO()
    sc.save(1234, buf)
    buf.seek(0)
    params, it = torch.load(buf, weights_only=False)
    assert it == 1234 and len(params) == 2 and len(params[0]) == 12                # gaussian_model.py:58-72
    assert params[0][1].shape == (50, 3) and params[0][2].shape == (50, 1, 3) and params[0][3].shape == (50, 15, 3)
    assert isinstance(params[0][10], dict) and "param_groups" in params[0][10] and params[0][11] == 10.0
    sc2 = training.GaussianScene([training.GaussianAsset(extent=10.0), training.GaussianAsset(extent=10.0)])
    sc2.restore(params, opt)
    for g, h in zip(sc.gaussians_assets, sc2.gaussians_assets):
        assert torch.equal(g._xyz, h._xyz) and torch.equal(g.optimizer.state[g._xyz]["exp_avg"], h.optimizer.state[h._xyz]["exp_avg"])


def test_ssim_and_range_frames():
    x = torch.rand(1, 24, 40)
    assert float(training.ssim(x, x)) == pytest.approx(1.0, abs=1e-6)
    # the separable matrix form equals the reference's 2-D convolution with the outer-product window (loss_utils.py:45-89)
    import torch.nn.functional as F
    y = torch.rand_like(x)
    g = torch.exp(-(torch.arange(11, dtype=torch.float32) - 5) ** 2 / (2 * 1.5 ** 2)); g = g / g.sum()
    win = (g[:, None] @ g[None, :]).expand(x.shape[0], 1, 11, 11).contiguous()
    conv = lambda a: F.conv2d(a, win, padding=5, groups=x.shape[0])
    mu1, mu2 = conv(x), conv(y)
    s1, s2, s12 = conv(x * x) - mu1 * mu1, conv(y * y) - mu2 * mu2, conv(x * y) - mu1 * mu2
    ref = (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))).mean()
    assert float(training.ssim(x, y)) == pytest.approx(float(ref), abs=2e-6)
    assert float(training.ssim(x, 1 - x)) < 0.5
    fr = training.RangeFrames()
    o = torch.zeros(2, 3, 3); d = torch.nn.functional.normalize(torch.rand(2, 3, 3), dim=-1)
    fr.add_frame(7, o, d, torch.full((2, 3), 5.0), torch.rand(2, 3), torch.tensor([[1, 0, 1], [1, 1, 0]]))
    pts = fr.inverse_projection_with_range(7, fr.get_depth(7), fr.get_mask(7))
    assert pts.shape == (4, 3) and torch.allclose(pts.norm(dim=1), torch.full((4,), 5.0), atol=1e-5)
    assert fr.train_frames == [7] and fr.get_range_rays(7)[1] is d


def test_range_ray_grid_matches_the_reference_sensor():
    """RangeFrames.range_rays against the golden grids produced by the reference's LiDARSensor.get_range_rays
    (tests/golden/rays_*.npz: KITTI bounds mode, a posed KITTI sensor, Waymo per-beam table with yaw and half-pixel offsets)."""
    import math, os
    gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    kb = [math.radians(-24.9), math.radians(2.0)]
    g = np.load(os.path.join(gd, "rays_kitti_16x256.npz"))
    o, d = training.RangeFrames.range_rays(16, 256, kb, torch.eye(4))
    np.testing.assert_allclose(d.numpy(), g["ray_d"], atol=2e-6); np.testing.assert_array_equal(o.numpy(), g["ray_o"])
    g = np.load(os.path.join(gd, "rays_kitti_posed_8x32.npz"))
    o, d = training.RangeFrames.range_rays(8, 32, kb, torch.from_numpy(g["sensor2world"]))
    np.testing.assert_allclose(d.numpy(), g["ray_d"], atol=2e-6); np.testing.assert_allclose(o.numpy(), g["ray_o"], atol=1e-7)
    g = np.load(os.path.join(gd, "rays_waymo_8x40.npz"))
    s2w = torch.from_numpy(g["sensor2world"])
    o, d = training.RangeFrames.range_rays(8, 40, g["beam_inclinations"].tolist(), s2w, "Waymo", sensor2ego=s2w)
    np.testing.assert_allclose(d.numpy(), g["ray_d"], atol=2e-6); np.testing.assert_allclose(o.numpy(), g["ray_o"], atol=1e-7)
    fr = training.RangeFrames()
    fr.add_range_image(3, torch.full((8, 40), 7.0), torch.zeros(8, 40), torch.ones(8, 40), g["beam_inclinations"].tolist(), s2w, "Waymo", s2w)
    pts = fr.inverse_projection_with_range(3, fr.get_depth(3))
    assert pts.shape == (320, 3) and torch.allclose((pts - s2w[:3, 3]).norm(dim=1), torch.full((320,), 7.0), atol=1e-4)


@pytest.mark.gpu
def test_short_optimisation_run_on_the_gpu():
    """Targets rendered from a ground-truth scene; a perturbed copy is optimised for 60 iterations through
    raytracing (fused pre-processing + tracer) + Chamfer + Adam + densification statistics: the loss must drop."""
    dev = torch.device("cuda:0")
    sc = scenes.make_scene(8000, seed=21, radius_scale=0.25)
    o, d = scenes.kitti_rays(16, 256)
    t = lambda a: torch.as_tensor(a, device=dev)
    def asset(noise):
        r = np.random.default_rng(1)
        op = sc["opacities"]
        a = training.GaussianAsset.from_tensors(
            t(sc["means"] + noise * r.normal(size=sc["means"].shape).astype(np.float32)), t(sc["shs"][:, :1]), t(sc["shs"][:, 1:]),
            t(np.log(sc["scales"])), t(sc["rotations"]), t(np.log(op / (1 - op)) - 3.0 * float(noise > 0)), extent=15.0)
        a.active_sh_degree = 3
        return a
    opt = training.default_options()
    opt.position_lr_init, opt.position_lr_final = 0.002, 0.0002
    bg = torch.tensor([0.0, 0.0, 1.0], device=dev)
    frames = training.RangeFrames()
    truth = training.GaussianScene([asset(0.0)])
    from lidar_rt_amd.renderer import raytracing
    args = types.SimpleNamespace(dynamic=False, opt=opt, pipe=types.SimpleNamespace())
    with torch.no_grad():
        pk = raytracing(0, truth.gaussians_assets, (t(o), t(d), torch.zeros(3, device=dev)), bg, args)
    mask = pk["raydrop"].squeeze(-1) < 0.6
    frames.add_frame(0, t(o), t(d), pk["depth"].squeeze(-1).detach(), pk["intensity"].squeeze(-1).detach(), mask)
    scene = training.GaussianScene([asset(0.05)])
    scene.training_setup(opt)
    hist = [training.training_step(scene, frames, 0, it, opt, bg) for it in range(1, 61)]
    first, last = float(torch.stack([h["loss"] for h in hist[:5]]).mean()), float(torch.stack([h["loss"] for h in hist[-5:]]).mean())
    assert np.isfinite(last) and last < 0.7 * first, (first, last)
    g = scene.gaussians_assets[0]
    assert float(g.denom.sum()) > 0 and float(g.xyz_gradient_accum.sum()) > 0       # densification statistics were fed
    # densify + prune on the live optimiser, then keep training
    opt.densify_from_iter, opt.densification_interval = 0, 1
    info = training.training_step(scene, frames, 0, 61, opt, bg)
    assert info["points"] == g._xyz.shape[0] and sum(info["densify"]) >= 0
    training.training_step(scene, frames, 0, 62, opt, bg)


@pytest.mark.gpu
def test_initialisation_from_points_uses_the_knn_operator():
    dev = torch.device("cuda:0")
    pts = torch.as_tensor(np.random.default_rng(0).normal(size=(3000, 3)).astype(np.float32) * 10, device=dev)
    a = training.GaussianAsset.from_points(pts, torch.rand(3000, 1, device=dev), extent=20.0)
    from simple_knn._C import distCUDA2
    want = torch.log(torch.sqrt(torch.clamp_min(distCUDA2(pts), 1e-7)))
    assert torch.equal(a._scaling[:, 0].detach(), want) and a._scaling.shape == (3000, 2)
    assert a._features_dc.shape == (3000, 1, 3) and a._features_rest.shape == (3000, 15, 3)
    assert float(a.get_opacity.detach().mean()) == pytest.approx(0.1, abs=1e-5)
    b = training.GaussianAsset.from_points(pts, torch.rand(3000, 1, device=dev), normals=torch.nn.functional.normalize(pts, dim=1), extent=20.0)
    R = training._rotation_matrix(b._rotation.detach())
    assert torch.allclose(R[:, :, 2], torch.nn.functional.normalize(pts, dim=1), atol=1e-4)   # third axis = normal
