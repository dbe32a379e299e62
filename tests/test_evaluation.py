"""lidar_rt_amd.evaluation: the metric set of the reference's eval.py:282-365 against numpy restatements (CPU), and the batched
forward-only rendering loop (GPU): frames are enqueued without a per-frame wait, perfect ground truth gives perfect scores."""
import time
import types

import numpy as np
import pytest
import torch

from lidar_rt_amd import evaluation, scenes


def test_image_and_raydrop_metrics_match_numpy_restatements():
    rng = np.random.default_rng(0)
    gt = rng.uniform(0, 90, (32, 64)).astype(np.float32); pred = (gt + rng.normal(0, 0.5, gt.shape)).astype(np.float32)
    m = evaluation.depth_metrics(torch.tensor(gt), torch.tensor(pred))
    g, p = np.clip(gt, 1e-6, 80), np.clip(pred, 1e-6, 80)                       # eval.py:282-297
    assert abs(float(m["rmse"]) - np.sqrt(((g - p) ** 2).mean())) < 1e-5
    assert abs(float(m["mae"]) - np.abs(g - p).mean()) < 1e-5
    assert abs(float(m["medae"]) - np.median(np.abs(g - p))) < 1e-6
    assert abs(float(m["psnr"]) - 10 * np.log10(80 ** 2 / ((p - g) ** 2).mean())) < 1e-3
    # SSIM: a scipy restatement of skimage.metrics.structural_similarity's defaults (7x7 uniform filter, sample covariance,
    # border cropped), which is what eval.py:299-301 calls with data_range = max(gt) - min(gt)
    from scipy.ndimage import uniform_filter
    def sk_ssim(x, y, R, win=7):
        x = x.astype(np.float64); y = y.astype(np.float64)
        f = lambda a: uniform_filter(a, size=win, mode="reflect")
        ux, uy = f(x), f(y); n = win * win / (win * win - 1.0)
        vx, vy, vxy = n * (f(x * x) - ux * ux), n * (f(y * y) - uy * uy), n * (f(x * y) - ux * uy)
        C1, C2 = (0.01 * R) ** 2, (0.03 * R) ** 2
        S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
        pad = (win - 1) // 2
        return S[pad:-pad, pad:-pad].mean()
    assert abs(float(m["ssim"]) - sk_ssim(p, g, g.max() - g.min())) < 1e-5
    assert float(evaluation.depth_metrics(torch.tensor(gt), torch.tensor(gt))["ssim"]) > 0.999
    gd = (rng.uniform(size=2000) < 0.3).astype(np.float32); pd = np.where(rng.uniform(size=2000) < 0.9, gd, 1 - gd).astype(np.float32)
    r = evaluation.raydrop_metrics(torch.tensor(gd), torch.tensor(pd))            # eval.py:333-349
    TP = ((gd == 1) & (pd == 1)).sum(); FP = ((gd == 0) & (pd == 1)).sum(); FN = ((gd == 1) & (pd == 0)).sum()
    prec, rec = TP / (TP + FP), TP / (TP + FN)
    assert abs(float(r["f1"]) - 2 * prec * rec / (prec + rec)) < 1e-6 and abs(float(r["acc"]) - (gd == pd).mean()) < 1e-6
    assert abs(float(r["rmse"]) - np.sqrt(((gd - pd) ** 2).mean())) < 1e-6


@pytest.mark.gpu
def test_batched_evaluation_of_a_scene_against_its_own_rendering():
    from lidar_rt_amd import training, renderer
    DEV = torch.device("cuda:0")
    sc = scenes.make_scene(40_000, seed=3, radius_scale=0.4)
    t = lambda a: torch.as_tensor(a, device=DEV)
    asset = training.GaussianAsset.from_tensors(t(sc["means"]), t(sc["shs"][:, :1]).contiguous(), t(sc["shs"][:, 1:]).contiguous(),
                                                torch.log(t(sc["scales"])), t(sc["rotations"]), training.inverse_sigmoid(t(sc["opacities"])),
                                                max_sh_degree=3, extent=30.0)
    asset.active_sh_degree = 3
    frames = training.RangeFrames()
    bg = torch.tensor(scenes.BG_DEFAULT, device=DEV)
    renderer.tracer_2dgs = None
    H, W = 32, 256
    rays = {}
    for f in range(6):
        o, d = scenes.range_rays(H, W, (np.radians(-24.9), np.radians(2.0)), scenes.pose_matrix((0.2 * f, 0.0, 0.0), yaw=0.02 * f), "KITTI")
        rays[f] = (t(o), t(d))
        frames.add_frame(f, rays[f][0], rays[f][1], torch.zeros(H, W, device=DEV), torch.zeros(H, W, device=DEV), torch.ones(H, W, device=DEV))
    first = evaluation.render_frames([asset], frames, range(6), bg)
    thr = float(torch.cat([first[f]["raydrop"].flatten() for f in range(6)]).median())   # a ray-drop threshold that splits this synthetic scene's rays
    for f in range(6):                                                            # ground truth = the scene's own rendering, masked like eval.py:224, :238
        hit = first[f]["raydrop"].squeeze(-1) < thr
        frames.add_frame(f, rays[f][0], rays[f][1], first[f]["depth"].squeeze(-1) * hit, first[f]["intensity"].squeeze(-1).clamp(0, 1) * hit, hit)
    res = evaluation.evaluate([asset], frames, list(range(6)), bg, raydrop_ratio=thr)
    assert res["mean"]["depth"]["rmse"] < 1e-5 and res["mean"]["intensity"]["rmse"] < 1e-6
    assert res["mean"]["raydrop"]["acc"] == 1.0 and res["mean"]["points"]["fscore"] > 0.999 and res["mean"]["points"]["chamfer_dist"] < 1e-8
    assert res["mean"]["depth"]["ssim"] > 0.999
    # the frames of a batch are enqueued without waiting for each other: behind a busy GPU the loop returns at once
    torch.cuda._sleep(1_000_000); torch.cuda.synchronize()
    t0 = time.perf_counter(); torch.cuda._sleep(20_000_000); torch.cuda.synchronize(); rate = 20_000_000 / (time.perf_counter() - t0)
    torch.cuda._sleep(int(0.3 * rate))
    marker = torch.cuda.Event(); marker.record()
    t0 = time.perf_counter()
    out = evaluation.render_frames([asset], frames, range(6), bg, check=False)
    host = time.perf_counter() - t0
    assert not marker.query() and host < 0.1, f"{host * 1e3:.1f} ms to enqueue 6 frames behind a busy GPU"
    renderer.tracer_2dgs.check()
    assert torch.equal(out[3]["depth"], first[3]["depth"])
    renderer.tracer_2dgs = None
