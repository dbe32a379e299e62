"""``python -m lidar_rt_amd.evaluate --data DIR --ckpt CKPT [--frames test|train|all]``: the evaluation loop of zju3dv/LiDAR-RT's ``eval.py``
(:370-470) on a file-backed sequence (``lidar_rt_amd.sequence``) and a checkpoint in the reference's layout (what ``python -m lidar_rt_amd.train``
writes: ``torch.save(([12-tuple per asset], iteration), path)``, gaussian_model.py:58-72).

Renders the chosen frames forward-only through ``renderer.raytracing`` (no host synchronisation between frames: ``evaluation.render_frames``),
computes the metric set of eval.py:282-365 (depth / intensity: rmse, mae, medae, ssim, psnr; ray-drop: rmse, accuracy, F1; points: Chamfer distance
and F-score through the package's own Chamfer operator) and prints ONE JSON object ``{"iteration", "frames", "mean", "per_frame"}``; ``--out`` also
writes it to a file.  Not reproduced (``evaluation.py``): LPIPS, the U-Net ray-drop refinement, image / point-cloud dumps.
"""
from __future__ import annotations

import argparse
import json
import sys
from types import SimpleNamespace

import torch


def run(args) -> dict:
    from . import evaluation, sequence, training
    if not torch.cuda.is_available():
        raise SystemExit("lidar_rt_amd.evaluate needs a HIP device (there is no CPU path)")
    dev = torch.device("cuda", args.device)
    torch.cuda.set_device(dev)
    seq = sequence.load_sequence(args.data, dev)
    opt = training.default_options()
    scene = sequence.scene_from_sequence(seq, max_points=args.max_points, seed=args.seed)
    scene.training_setup(opt)
    model_params, iteration = torch.load(args.ckpt, map_location=dev, weights_only=False)
    scene.restore(model_params, opt)
    frames = {"test": seq.test_frames or seq.train_frames, "train": seq.train_frames, "all": sorted(set(seq.train_frames) | set(seq.test_frames))}[args.frames]
    if args.max_frames > 0:
        frames = frames[:args.max_frames]
    bg = torch.tensor([0.0, 0.0, 1.0], device=dev)          # the reference's background for (intensity, ray-hit, ray-drop): eval.py:104
    rargs = SimpleNamespace(dynamic=bool(seq.meta.get("dynamic")), opt=SimpleNamespace(use_rayhit=bool(getattr(opt, "use_rayhit", False))), pipe=SimpleNamespace())
    res = evaluation.evaluate(scene.gaussians_assets, seq.frames, frames, bg, rargs, raydrop_ratio=args.raydrop_ratio, use_gt_mask=args.use_gt_mask,
                              max_depth=args.max_depth)
    return {"iteration": int(iteration), "frames": [int(f) for f in frames], "mean": res["mean"], "per_frame": {str(k): v for k, v in res["frames"].items()}}


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m lidar_rt_amd.evaluate", description=__doc__.split("\n\n")[0])
    ap.add_argument("--data", required=True, help="sequence directory (lidar_rt_amd.sequence layout)")
    ap.add_argument("--ckpt", required=True, help="checkpoint in the reference's (model_params, iteration) layout (python -m lidar_rt_amd.train writes them)")
    ap.add_argument("--frames", choices=("test", "train", "all"), default="test", help="which frames (test falls back to train when the sequence names no test frames)")
    ap.add_argument("--max-frames", type=int, default=0)
    ap.add_argument("--raydrop-ratio", type=float, default=0.4, help="eval.py:72")
    ap.add_argument("--use-gt-mask", action="store_true", help="mask the renders with the ground-truth ray-hit mask instead of the predicted one (eval.py:184)")
    ap.add_argument("--max-depth", type=float, default=80.0)
    ap.add_argument("--max-points", type=int, default=2_000_000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out", default=None, help="also write the JSON object to this file")
    args = ap.parse_args(argv)
    res = run(args)
    txt = json.dumps(res)
    print(txt, flush=True)
    if args.out:
        with open(args.out, "w") as f:
            f.write(txt + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
