"""``python -m lidar_rt_amd.train --data DIR --iters N [--gpus N]``: the training loop of zju3dv/LiDAR-RT's ``train.py`` (:67-447) on a
file-backed sequence (``lidar_rt_amd.sequence``: range images + poses + tracking boxes in a neutral on-disk layout).

Per iteration (train.py:125-220): pick a training frame, ``training_step`` (render through ``renderer.raytracing`` -> losses -> backward ->
``scene.optimize``: Adam, densification, pruning, opacity reset), log, and every ``--save-every`` iterations write a checkpoint in the
REFERENCE's layout -- ``torch.save(([12-tuple per asset], iteration), DIR_OUT/chkpnt<iteration>.pth)`` (gaussian_model.py:58-72,
train.py:229-232) -- which ``--resume`` reads back.  ``--gpus N`` starts one process per GPU (torch.distributed over RCCL); the frame is
sharded by azimuth sector and the gradients are exchanged (``lidar_rt_amd.parallel.ShardedTracer`` behind ``renderer.sharded``): every rank
holds the full gradient and runs the same optimizer step.

What a resumed run reproduces: the parameters, the Adam moments, the densification statistics and the iteration come from the checkpoint
bit for bit; the frame order and the random draws of the densification are functions of (seed, iteration), not of process state.  The
step's float sums themselves depend on the arrival order of integer atomics inside the backward (like the reference's float atomics), so
two runs of the same iterations agree to rounding, not bit for bit.
"""
from __future__ import annotations

import argparse
import json
import os
import random
import subprocess
import sys
import time

import torch


def frame_of(seed: int, iteration: int, frames):
    """The training frame of an iteration: a function of (seed, iteration) only, so that a resumed run sees the same sequence."""
    return frames[random.Random(seed * 1_000_003 + iteration).randrange(len(frames))]


def run(args) -> dict:
    from . import renderer, sequence, training
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("lidar_rt_amd.train needs a HIP device (there is no CPU path)")
    single_dev = os.environ.get("LRT_SINGLE_DEVICE", "0") == "1"            # developer switch: all ranks on cuda:0 (tests on a one-GPU box)
    dev = torch.device("cuda", 0 if single_dev else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("LRT_DIST_BACKEND", "nccl")
        dist.init_process_group(backend=backend, **({"device_id": dev} if backend == "nccl" else {}))
        from .parallel import ShardedTracer
        renderer.sharded = ShardedTracer(exchange="sparse", deferred_accum=not args.exact_accum, deterministic=args.deterministic)
    renderer.deferred_accum = not args.exact_accum       # the loop reads the hit weights after loss.backward() only (train.py:156,219)
    renderer.deterministic = bool(args.deterministic)    # gradient sums in a fixed order, a forward without learnt state: a resumed run repeats the uninterrupted one bit for bit

    seq = sequence.load_sequence(args.data, dev)
    opt = training.default_options()
    for kv in args.opt:
        k, v = kv.split("=", 1)
        if not hasattr(opt, k):
            raise SystemExit(f"--opt {k}: not an option of lidar_rt_amd.training.default_options()")
        setattr(opt, k, type(getattr(opt, k))(float(v)) if not isinstance(getattr(opt, k), bool) else v.lower() in ("1", "true"))
    opt.iterations = max(opt.iterations, args.iters)
    torch.manual_seed(args.seed)
    scene = sequence.scene_from_sequence(seq, max_points=args.max_points, seed=args.seed)
    scene.training_setup(opt)
    first = 1
    if args.resume:
        model_params, it0 = torch.load(args.resume, map_location=dev, weights_only=False)
        scene.restore(model_params, opt)
        first = int(it0) + 1
    os.makedirs(args.out, exist_ok=True)
    bg = torch.tensor([0.0, 0.0, 1.0], device=dev)       # the reference's background for (intensity, ray-hit, ray-drop): train.py:106
    log, t0 = [], time.perf_counter()
    res = {}
    for it in range(first, args.iters + 1):
        torch.manual_seed(args.seed * 1_000_003 + it)      # the densification's random draws: a function of (seed, iteration) on every rank
        frame = frame_of(args.seed, it, seq.train_frames)
        res = training.training_step(scene, seq.frames, frame, it, opt, bg, dynamic=bool(seq.meta.get("dynamic")))
        if it % args.log_every == 0 or it == args.iters:
            row = {"iteration": it, "frame": int(frame), "loss": float(res["loss"]), "depth": float(res["depth"]), "intensity": float(res["intensity"]),
                   "raydrop": float(res["raydrop"]), "points": int(res["points"]), "seconds": round(time.perf_counter() - t0, 3)}
            log.append(row)
            if rank == 0:
                print(json.dumps(row), flush=True)
        if rank == 0 and (it % args.save_every == 0 or it == args.iters):
            scene.save(it, os.path.join(args.out, f"chkpnt{it}.pth"))
    if renderer.sharded is not None:
        renderer.sharded.check(wait=True)
    elif renderer.tracer_2dgs is not None:
        renderer.tracer_2dgs.check(dev)
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier(); dist.destroy_process_group()
    return {"log": log, "scene": scene, "sequence": seq, "last": res}


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m lidar_rt_amd.train", description=__doc__.split("\n\n")[0])
    ap.add_argument("--data", required=True, help="sequence directory (lidar_rt_amd.sequence layout; tools/make_sequence.py writes synthetic ones)")
    ap.add_argument("--iters", type=int, default=30_000)
    ap.add_argument("--out", default=None, help="checkpoint / log directory (default: DATA/output)")
    ap.add_argument("--gpus", type=int, default=1, help="N > 1: one process per GPU, azimuth-sharded frames, replicated optimizer")
    ap.add_argument("--resume", default=None, help="checkpoint to continue from (the reference's (model_params, iteration) tuple)")
    ap.add_argument("--save-every", type=int, default=1000)
    ap.add_argument("--log-every", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-points", type=int, default=2_000_000, help="cap of the background's initial point cloud")
    ap.add_argument("--opt", action="append", default=[], help="training option name=value (lidar_rt_amd.training.default_options)")
    ap.add_argument("--exact-accum", action="store_true", help="hit weights complete at the forward (the reference's contract) instead of written by "
                    "the backward (renderer.deferred_accum, the default here: the loop reads them after the backward only)")
    ap.add_argument("--deterministic", action="store_true", help="bit-reproducible steps (Tracer(deterministic=True): the backward adds a Gaussian's records up by ray and "
                    "the pieces of long runs in order, the forward keeps no learnt tables): ~1.4 x the tracer time; a run resumed from a checkpoint then equals the "
                    "uninterrupted one bit for bit (with lambda_cd = 0: the Chamfer backward adds with float atomics)")
    args = ap.parse_args(argv)
    if args.exact_accum and args.deterministic:
        ap.error("--deterministic takes the hit weights from the backward (the forward's are float atomics): not with --exact-accum")
    if args.out is None:
        args.out = os.path.join(args.data, "output")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # one rank per GPU under torch.distributed.run, like bench.py --gpus N
        import socket
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        if torch.cuda.device_count() < args.gpus and os.environ.get("LRT_SINGLE_DEVICE", "0") != "1":
            print(f"[train] refusing: --gpus {args.gpus} but {torch.cuda.device_count()} visible devices", file=sys.stderr)
            return 2
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "-m", "lidar_rt_amd.train"] + (argv if argv is not None else sys.argv[1:])
        return subprocess.call(cmd)
    run(args)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
