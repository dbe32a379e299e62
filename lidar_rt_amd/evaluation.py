"""Evaluation-loop counterpart of the reference's ``eval.py`` for the operators of this package:

* ``render_frames``  -- ``Evaluator.record_render`` (eval.py:100-124): forward-only per-frame rendering through
  ``renderer.raytracing``.  The reference's tracer ends every call in ``cudaStreamSynchronize`` (DLT/trace_surfels.cpp:260); here
  the frames are ENQUEUED back to back (``Tracer.deferred_checks``), the overflow status of all of them is checked once at the end.
* the metric set of eval.py:282-365 on device tensors: depth / intensity (rmse, mae, medae, ssim = skimage's 7x7 uniform-window
  structural_similarity with the ground truth's data range, psnr), ray-drop (rmse, accuracy, F1 at ``raydrop_ratio`` = 0.4), points (Chamfer distance and F-score at 5 cm through the package's own ``chamfer_3DDist``).
  Not reproduced: LPIPS (needs pretrained network weights), the U-Net ray-drop refinement, image / point-cloud dumps.
* ``evaluate``       -- the loop of eval.py:370-470: per-frame metrics and their means.

The Gaussian assets and the sensor are duck-typed like in ``renderer.raytracing`` (``training.GaussianScene`` / ``RangeFrames``).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Iterable, List, Optional, Sequence

import torch

from . import renderer


def render_frames(gaussian_assets: Sequence, sensor, frame_ids: Iterable, background: torch.Tensor, args=None,
                  check: bool = True) -> Dict[object, Dict[str, torch.Tensor]]:
    """Forward-only renders of `frame_ids` (eval.py:124), no host synchronisation between frames.  Returns
    ``{frame: {"depth", "intensity", "raydrop"}}`` (each (H, W, 1), detached)."""
    if args is None:
        args = SimpleNamespace(dynamic=False, opt=SimpleNamespace(use_rayhit=False), pipe=SimpleNamespace())
    if renderer.tracer_2dgs is None:
        from .diff_lidar_tracer import Tracer
        renderer.tracer_2dgs = Tracer()
    tr = renderer.tracer_2dgs
    was_training, was_deferred = tr.training, tr.deferred_checks
    tr.eval(); tr.deferred_checks = True
    out = {}
    try:
        with torch.no_grad():
            for f in frame_ids:
                pkg = renderer.raytracing(f, gaussian_assets, sensor, background, args)
                out[f] = {k: pkg[k].detach() for k in ("depth", "intensity", "raydrop")}
        if check:
            tr.check()                                   # one wait for the whole batch; raises if any frame overflowed
    finally:
        tr.train(was_training); tr.deferred_checks = was_deferred
    return out


# ---------------------------------------------------------------------------------------------------------------- metrics
def ssim_uniform(pred: torch.Tensor, gt: torch.Tensor, data_range: torch.Tensor, win: int = 7) -> torch.Tensor:
    """`skimage.metrics.structural_similarity(pred, gt, data_range=...)` with its defaults, which is what eval.py:299-301 / :323-325
    calls: 7x7 UNIFORM window, K1 = 0.01, K2 = 0.03, sample covariance (normalised by N / (N - 1)), mean over the image with a
    border of (win - 1) / 2 pixels cropped -- on that interior the uniform filter equals an un-padded average pooling."""
    import torch.nn.functional as F
    x = pred.reshape(1, 1, *pred.shape[-2:]).double(); y = gt.reshape(1, 1, *gt.shape[-2:]).double()
    pool = lambda a: F.avg_pool2d(a, win, stride=1)
    ux, uy = pool(x), pool(y)
    norm = win * win / (win * win - 1.0)
    vx = norm * (pool(x * x) - ux * ux); vy = norm * (pool(y * y) - uy * uy); vxy = norm * (pool(x * y) - ux * uy)
    R = data_range.double()
    C1, C2 = (0.01 * R) ** 2, (0.03 * R) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2))
    return S.mean().float()


def _median(x: torch.Tensor) -> torch.Tensor:
    """numpy's median: the mean of the two middle elements for an even count (torch.median returns the lower one)."""
    v = x.flatten().sort().values
    n = v.numel()
    return 0.5 * (v[(n - 1) // 2] + v[n // 2])


def _image_metrics(gt: torch.Tensor, pred: torch.Tensor, lo: float, hi: float) -> Dict[str, torch.Tensor]:
    """eval.py:282-331: clamp both to [lo, hi]; rmse, mae, medae, ssim (skimage defaults, data range of the clamped ground truth),
    psnr with peak `hi`.  LPIPS is not reproduced (pretrained network weights)."""
    gt = gt.reshape(gt.shape[0], gt.shape[1]).clamp(lo, hi).float(); pred = pred.reshape(gt.shape).clamp(lo, hi).float()
    err = gt - pred
    mse = (err * err).mean()
    return {"rmse": mse.sqrt(), "mae": err.abs().mean(), "medae": _median(err.abs()),
            "ssim": ssim_uniform(pred, gt, gt.max() - gt.min()), "psnr": 10.0 * torch.log10(hi * hi / mse.clamp_min(1e-30))}


def depth_metrics(gt, pred, min_depth: float = 1e-6, max_depth: float = 80.0):
    return _image_metrics(gt, pred, min_depth, max_depth)


def intensity_metrics(gt, pred, min_intensity: float = 1e-6, max_intensity: float = 1.0):
    return _image_metrics(gt, pred, min_intensity, max_intensity)


def raydrop_metrics(gt_drop: torch.Tensor, pred_drop_mask: torch.Tensor) -> Dict[str, torch.Tensor]:
    """eval.py:333-349 on (1 - rayhit) masks: rmse, accuracy and F1 of the predicted drop mask."""
    gt = gt_drop.reshape(-1).float(); pr = pred_drop_mask.reshape(-1).float()
    tp = ((gt == 1) & (pr == 1)).sum().float(); fp = ((gt == 0) & (pr == 1)).sum().float(); fn = ((gt == 1) & (pr == 0)).sum().float()
    precision = tp / (tp + fp).clamp_min(1.0); recall = tp / (tp + fn).clamp_min(1.0)
    return {"rmse": ((gt - pr) ** 2).mean().sqrt(), "acc": (gt == pr).float().mean(),
            "f1": 2 * precision * recall / (precision + recall).clamp_min(1e-30)}


def points_metrics(gt_pts: torch.Tensor, pred_pts: torch.Tensor, threshold: float = 0.05) -> Dict[str, torch.Tensor]:
    """eval.py:351-365 + compute_fscore :270-280: Chamfer distance (sum of the two mean squared nearest distances) and the F-score
    of the squared distances at `threshold`, through the package's HIP Chamfer operator."""
    from .chamfer3D import chamfer_3DDist
    if gt_pts.shape[0] == 0 or pred_pts.shape[0] == 0:
        z = torch.zeros((), device=gt_pts.device)
        return {"chamfer_dist": z + float("nan"), "fscore": z}
    d1, d2, _, _ = chamfer_3DDist()(gt_pts[None].float().contiguous(), pred_pts[None].float().contiguous())
    return {"chamfer_dist": d1.mean() + d2.mean(), "fscore": fscore(d1, d2, threshold)[0]}


def fscore(dist1: torch.Tensor, dist2: torch.Tensor, threshold: float = 0.001):
    """eval.py:266-280 ``compute_fscore`` on squared nearest distances (B, N) / (B, M): (fscore, precision_1, precision_2), NaN -> 0."""
    p1 = (dist1 < threshold).float().mean(dim=-1); p2 = (dist2 < threshold).float().mean(dim=-1)
    f = 2 * p1 * p2 / (p1 + p2)
    return torch.nan_to_num(f, nan=0.0).reshape(-1)[0], p1.reshape(-1)[0], p2.reshape(-1)[0]


def evaluate(gaussian_assets: Sequence, sensor, frame_ids: Sequence, background: torch.Tensor, args=None,
             raydrop_ratio: float = 0.4, use_gt_mask: bool = False, max_depth: float = 80.0) -> Dict[str, object]:
    """Per-frame metrics and their means (eval.py:370-470; `raydrop_ratio` 0.4 as eval.py:72).  `sensor` offers get_depth /
    get_intensity / get_mask / inverse_projection_with_range like ``training.RangeFrames``.  As in ``record_render`` the rendered
    depth and the clamped rendered intensity are multiplied by the ray-hit mask (ground-truth or predicted, eval.py:184, :224, :238)
    before they are compared with the ground truth.  One device->host transfer at the end."""
    frame_ids = list(frame_ids)
    renders = render_frames(gaussian_assets, sensor, frame_ids, background, args)
    per_frame: Dict[object, Dict[str, Dict[str, torch.Tensor]]] = {}
    for f in frame_ids:
        r = renders[f]
        gt_hit = sensor.get_mask(f).bool()
        pred_hit = (r["raydrop"].squeeze(-1) < raydrop_ratio)
        mask = gt_hit if use_gt_mask else pred_hit
        gt_pts = sensor.inverse_projection_with_range(f, sensor.get_depth(f), gt_hit)
        pred_pts = sensor.inverse_projection_with_range(f, r["depth"].squeeze(-1), mask)
        mk = mask.to(r["depth"].dtype)
        per_frame[f] = {"depth": depth_metrics(sensor.get_depth(f), r["depth"].squeeze(-1) * mk, max_depth=max_depth),
                        "intensity": intensity_metrics(sensor.get_intensity(f).clamp(0, 1), r["intensity"].squeeze(-1).clamp(0, 1.0) * mk),
                        "raydrop": raydrop_metrics(1 - gt_hit.float(), 1 - pred_hit.float()),
                        "points": points_metrics(gt_pts, pred_pts)}
    # one transfer: stack every scalar
    keys = [(f, g, m) for f in frame_ids for g in per_frame[f] for m in per_frame[f][g]]
    vals = torch.stack([per_frame[f][g][m].float().reshape(()) for f, g, m in keys]).cpu().tolist() if keys else []
    out_frames: Dict[object, Dict[str, Dict[str, float]]] = {}
    for (f, g, m), v in zip(keys, vals):
        out_frames.setdefault(f, {}).setdefault(g, {})[m] = v
    mean: Dict[str, Dict[str, float]] = {}
    for g in ("depth", "intensity", "raydrop", "points"):
        for m in (per_frame[frame_ids[0]][g] if frame_ids else {}):
            xs = [out_frames[f][g][m] for f in frame_ids if out_frames[f][g][m] == out_frames[f][g][m]]
            mean.setdefault(g, {})[m] = sum(xs) / len(xs) if xs else float("nan")
    return {"frames": out_frames, "mean": mean}
