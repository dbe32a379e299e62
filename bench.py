#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native LiDAR Gaussian tracer.

Metric (BASELINE.json): LiDAR rays/s, forward + backward, 1 M Gaussians, 2048x64 sweep (workload S1M,
SURVEY.md section 8(d)); reported with the HBM-roofline fraction of the dominant kernel and with the CPU
oracle timed beside it.

One STEP = what the reference does once per training iteration for this operator
(lib/gaussian_renderer/__init__.py:142-160 + loss.backward()):
    acceleration-structure rebuild (fused build2DRectangle + LBVH)  ->  trace_surfels  ->  trace_surfels_backward
on inputs already resident in HBM.  With N > 1 the frame is sharded by azimuth sector (strong scaling: the
frame is fixed), slabs are all-gathered and the fused gradient buffer is all-reduced over RCCL.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL / cross-process tensors need on this driver

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from lidar_rt_amd import scenes  # noqa: E402

HBM_PEAK_BYTES_PER_S = 8.0e12        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3e12 achievable


def algorithmic_bytes(C: float, K: float, deg: int = 3):
    """SURVEY.md section 8(d) byte model, per ray: (forward, backward)."""
    n = 12 * (deg + 1) ** 2
    fwd = 24 + 36 + 40 * C + (n + 8) * K
    bwd = 24 + 36 + 36 + 40 * C + (n + 2 * (40 + n)) * K
    return fwd, bwd


def cpu_baseline(sc, ray_o, ray_d, deg, bg, dL, col_stride=8):
    """Oracle ("port") on all host cores, on every `col_stride`-th column of the same frame (whole frame when
    the strided probe finishes in < 2 s, so that the sample is ~10-30 s of CPU work on small and large hosts)."""
    from oracle import oracle
    ncores = oracle.num_threads()
    if col_stride > 1:
        probe = cpu_baseline(sc, ray_o[:, ::col_stride], ray_d[:, ::col_stride], deg, bg, dL[:, ::col_stride], 1)
        if probe["seconds"] >= 2.0:
            probe["sample"] = f"every {col_stride}th azimuth column of the frame: " + probe["sample"]
            return probe
        # fast host: repeat the whole frame until ~10 s of CPU work have been timed
        reps, tot_s, tot_rays, last = 0, 0.0, 0, None
        while tot_s < 10.0 and reps < 12:
            last = cpu_baseline(sc, ray_o, ray_d, deg, bg, dL, 1)
            tot_s += last["seconds"]; tot_rays += ray_o.shape[0] * ray_o.shape[1]; reps += 1
        last["value"] = tot_rays / tot_s; last["seconds"] = tot_s
        last["sample"] = f"whole frame x{reps} ({tot_s:.1f} s of CPU work); last repetition: " + last["sample"]
        return last
    o = np.ascontiguousarray(ray_o[:, ::col_stride]); d = np.ascontiguousarray(ray_d[:, ::col_stride])
    g = np.ascontiguousarray(dL[:, ::col_stride])
    t0 = time.time()
    orc = oracle.Oracle(sc["means"], sc["scales"], sc["rotations"], sc["opacities"], "f32")
    t1 = time.time()
    fw = orc.forward(o, d, sc["shs"], deg, bg)
    t2 = time.time()
    orc.backward(o, d, sc["shs"], deg, bg, fw["out"], g)
    t3 = time.time()
    n = o.shape[0] * o.shape[1]
    phys, logical = _cpu_counts()
    return {"value": n / (t3 - t1), "unit": "rays/s", "cores": phys, "threads": ncores, "logical_cpus": logical, "kind": "port", "seconds": t3 - t1,
            "value_with_build": n / (t3 - t0), "build_seconds": t1 - t0,
            "sample": f"{o.shape[0]}x{o.shape[1]} = {n} rays, forward+backward, OpenMP over rays on {ncores} threads "
                      f"({phys} physical cores; fwd {t2 - t1:.2f}s, bwd {t3 - t2:.2f}s; the oracle's single-threaded CPU BVH build, {t1 - t0:.2f}s, "
                      f"is not in `value` -- `value_with_build` includes it, as the GPU step includes its LBVH build)",
            "cpu_model": _cpu_model()}


def _cpu_counts():
    """(physical cores, logical CPUs) of the host."""
    logical = os.cpu_count() or 1
    try:
        cores = set(); phys_id = core_id = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"): phys_id = line.split(":")[1].strip()
                elif line.startswith("core id"): core_id = line.split(":")[1].strip()
                elif not line.strip():
                    if phys_id is not None and core_id is not None: cores.add((phys_id, core_id))
                    phys_id = core_id = None
        if cores:
            return len(cores), logical
    except OSError:
        pass
    return logical, logical


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return int(sk.getsockname()[1])


def self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no torch.distributed environment: start the N ranks here -- the command the driver would
    have used (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`), one device per rank -- instead
    of silently measuring one rank.  Refuses (exit code 2) when the node has fewer than N devices (LRT_SINGLE_DEVICE=1, the developer
    switch that stacks all ranks on cuda:0, lifts that)."""
    n = int(args.gpus)
    single = os.environ.get("LRT_SINGLE_DEVICE", "0") == "1"
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--print-launch"]
    if args.print_launch:
        print(json.dumps({"launch": cmd, "devices_visible": have, "single_device": single}), flush=True)
        return 0
    if have < (1 if single else n):
        print(f"[bench] refusing: --gpus {n} needs {n} visible HIP devices, this node shows {have} "
              "(no CPU fallback; LRT_SINGLE_DEVICE=1 LRT_DIST_BACKEND=gloo stacks the ranks on one device for development)", file=sys.stderr)
        return 2
    import subprocess
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0"); env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="s1m", choices=["s1m", "s10k", "s200k", "waymo4m"], help="s1m = the headline (BASELINE configs[1]); waymo4m = the shape of "
                    "configs[4] (4 M Gaussians, 64x2650 Waymo-style grid, ~4.5 ms step on one GPU): the size at which an 8-way azimuth split has enough work per rank")
    ap.add_argument("--opt", action="append", default=[], help="library option name=value (e.g. fwd_mode=0, bwd_mode=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange", default="sparse", choices=["owner", "dense", "sparse", "auto"], help="gradient exchange for N > 1: sparse (default) = "
                    "all_gather of the touched Gaussians' rows, every rank ends with the full gradient (what a replicated optimizer -- the sharded "
                    "training step -- consumes; nothing else has to be synchronised); dense = all_reduce of the flat buffer (replicated); owner = "
                    "every Gaussian's gradient is reduced to its owning rank only (reduce-scatter semantics: NOT a complete training exchange, "
                    "parameters and Adam moments of the touched rows would still have to be synchronised)")
    ap.add_argument("--min-seconds", type=float, default=3.0, help="length of the SUSTAINED window that runs in front of the timed one (whole blocks of --steps until it "
                    "lasts at least this long; clocks and caches are warm afterwards, and driver-side telemetry sees the GPU busy).  It is reported as `sustained` "
                    "BESIDE `value`; `value` itself always comes from EXACTLY --steps steps, bracketed by barrier + synchronize.  0 = no sustained window")
    ap.add_argument("--check-sum", action="store_true", help="add checksums of the (all-gathered / all-reduced) results")
    ap.add_argument("--via", default="direct", choices=["direct", "tracer"], help="direct (default): the step drives the `_C` binding through ShardedTracer "
                    "with preallocated gradient views; tracer: the DROP-IN path a maintainer gets -- diff_lidar_tracer.Tracer -> torch.autograd -> `_C` "
                    "(fresh output / gradient tensors per call, the three dead zero outputs), N = 1 only.  The line always carries the other path's "
                    "rays/s as `drop_in_path` / `direct_path` when --both-paths is given")
    ap.add_argument("--both-paths", dest="both_paths", action="store_true", default=True,
                    help="(default on one GPU) time the other --via path too, same window length, and report it beside `value` as `drop_in_path` / `direct_path`")
    ap.add_argument("--no-both-paths", dest="both_paths", action="store_false", help="time only the --via path")
    ap.add_argument("--no-vary", dest="vary", action="store_false", help="skip the `value_varying` window")
    ap.add_argument("--no-deferred", action="store_true", help="skip the `value_deferred_accum` window (the training configuration: hit weights written by the backward)")
    ap.add_argument("--pose-inside", action="store_true", help="--vary poses 0.1 m from clutter instead of inside the scene's empty cylinder: hundreds of rays start closer than "
                    "0.2 m to a quad and take the literal K-buffer replay (k_fwd_near); reported as `value_varying.near_rays_per_frame` / `near_us`")
    ap.add_argument("--print-launch", action="store_true", help="with --gpus N > 1 and no torch.distributed environment: print the launch command as JSON and exit (tests)")
    ap.add_argument("--vary", action="store_true", default=True, help="(default on one GPU) a second timed window in which the frame CHANGES from step to step like in training "
                    "(train.py:125-148 draws another frame per iteration and the optimizer moves the parameters): 4 sensor poses (yawed / translated ray sets) "
                    "in rotation and an Adam-sized perturbation of means / scales / opacities between steps.  Reported as `value_varying` BESIDE `value`: "
                    "it shows what the temporal speculation (per-tile learned slab widths, Morton box of the previous build, speculated hit counts) is worth "
                    "when consecutive frames differ")
    ap.add_argument("--graph", action="store_true", help="library option graph=1: every API call's launch sequence is replayed from an instantiated HIP graph (one "
                    "graph launch per lrt_build / lrt_forward / lrt_backward).  The step then runs on a side stream (the legacy default stream cannot be captured) and "
                    "without the library's HIP-event timers (phase_ms / roofline per-kernel times are null).  What it buys is host launch time: S10k is launch-bound")
    ap.add_argument("--no-build-in-step", action="store_true", help="exclude the LBVH rebuild from the step")
    ap.add_argument("--time-every", type=int, default=8, help="the library's region timers record their HIP events on every K-th step of the timed window (1 = every step)")
    ap.add_argument("--no-stats-step", action="store_true", help="skip the one instrumented (untimed) step that collects the traversal counters: the profiling "
                    "scripts use it so that every kernel of the trace is a kernel of a regular step (the counters' k_fwd_cr4<.., true> instantiation is not)")
    ap.add_argument("--refit-every", type=int, default=0, help="K > 0: K lrt_refit calls between full LBVH builds (NOT the headline "
                    "configuration: the reference rebuilds its acceleration structure on every call, and so does the default step)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))                      # re-enters this file once per rank under torch.distributed.run
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print(f"[bench] refusing: --gpus {args.gpus} but torch.distributed started WORLD_SIZE={world} ranks; the line would report "
                  f"{world} rank(s) as {args.gpus}", file=sys.stderr)
        raise SystemExit(2)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback in the product path)")
    # developer switches to exercise the N>1 path on a 1-GPU box: all ranks on cuda:0, gloo instead of RCCL
    single_dev = os.environ.get("LRT_SINGLE_DEVICE", "0") == "1"
    backend = os.environ.get("LRT_DIST_BACKEND", "nccl")
    if world > 1 and not single_dev and torch.cuda.device_count() < world:
        raise SystemExit(f"[bench] refusing: {world} ranks but {torch.cuda.device_count()} visible devices (one device per rank)")
    dev_index = 0 if single_dev else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    # ---------------- synthetic workload (identical on every rank: seeded)
    if args.workload == "s1m":
        sc, ro, rd = scenes.s1m(); wl = "S1M: 1,000,000 Gaussians, 64x2048 KITTI-style sweep (BASELINE configs[1])"
    elif args.workload == "waymo4m":
        sc, ro, rd = scenes.waymo_dynamic_4m(); wl = "Waymo-dynamic shape: 4,000,000 Gaussians (background + posed actors), 64x2650 grid (BASELINE configs[4])"
    elif args.workload == "s200k":
        sc = scenes.make_scene(200_000, radius_scale=0.5); ro, rd = scenes.kitti_rays(32, 512); wl = "S200k (dev)"
    else:
        sc, ro, rd = scenes.s10k(); wl = "S10k: 10,000 Gaussians, 16x256 rays (BASELINE configs[0])"
    H, W = ro.shape[:2]
    deg = 3
    bg_np = scenes.BG_DEFAULT
    dL_np = scenes.upstream_grad(H, W)
    t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
    ray_o, ray_d = torch.as_tensor(ro, device=dev), torch.as_tensor(rd, device=dev)
    bg = torch.as_tensor(bg_np, device=dev)
    dL = torch.as_tensor(dL_np, device=dev)

    from lidar_rt_amd.parallel import ShardedTracer, column_slab
    tr = ShardedTracer(exchange=args.exchange)
    st = tr.backend.state
    for kv in args.opt:
        k_, v_ = kv.split("=")
        st.set_option(k_, int(v_))
    st.refit_interval = max(args.refit_every, 0)
    if args.graph:
        st.set_option("graph", 1)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.set_stream(side)                                   # everything below is enqueued on the side stream

    def step_direct():
        out, _ = tr.forward(ray_o, ray_d, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], deg, bg,
                            rebuild=not args.no_build_in_step, cull_key="bench-frame")     # the ray set's name: sizes a rank's culled build without a read-back
        g = tr.backward(t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], deg, bg, dL)
        return out, g

    # the drop-in path: what lib/gaussian_renderer/__init__.py:142-160 + loss.backward() execute per iteration
    from lidar_rt_amd.diff_lidar_tracer import Tracer, TracingSettings, _C as _binding
    drop = {"tracer": None}

    def step_tracer():
        if drop["tracer"] is None:
            trc = Tracer(); trc.optix_context = st                      # the same state object (options, timing, statistics)
            e_ = torch.empty(0, device=dev)
            drop["tracer"] = (trc, TracingSettings(None, None, None, None, bg, 1.0, e_, e_, deg, torch.zeros(3, device=dev), False, False),
                              {k: v.detach().clone().requires_grad_(True) for k, v in t.items()})
        trc, ts_, leaf = drop["tracer"]
        for v in leaf.values():
            v.grad = None
        if not args.no_build_in_step:
            trc.build_from_gaussians(leaf["means"], leaf["scales"], leaf["rotations"], leaf["opacities"])
        out, acc = trc(ray_o, ray_d, None, leaf["means"], torch.zeros_like(leaf["means"]), shs=leaf["shs"], opacities=leaf["opacities"],
                       scales=leaf["scales"], rotations=leaf["rotations"], tracer_settings=ts_)
        out.backward(dL)
        return out.detach(), {"means": leaf["means"].grad, "shs": leaf["shs"].grad, "accum": acc}

    if args.via == "tracer" and world > 1:
        raise SystemExit("--via tracer is the single-GPU drop-in path; N > 1 runs through ShardedTracer (renderer.sharded for training)")
    step = step_tracer if args.via == "tracer" else step_direct

    if args.no_build_in_step:
        tr.backend.build(t["means"], t["scales"], t["rotations"], t["opacities"])
    for _ in range(max(args.warmup, 0)):
        step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the SUSTAINED window first: whole blocks of --steps until it lasts --min-seconds (one calibration block sizes it; every rank uses
    # the same count).  It is reported beside `value` and leaves clocks / caches / the library's speculation warm for the timed window.
    st.set_option("timing_every", max(1, args.time_every))
    reps, sustained = 0, None
    if args.min_seconds > 0 and args.steps > 0:
        barrier(); tc = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier(); cal = time.perf_counter() - tc
        reps = max(1, int(np.ceil(args.min_seconds / max(cal, 1e-6))))
        if world > 1:
            tr_ = torch.tensor([reps], dtype=torch.int64, device=dev); dist.all_reduce(tr_, op=dist.ReduceOp.MAX); reps = int(tr_.item())
        st.enable_timing(not args.graph)
        barrier(); ts0 = time.perf_counter()
        for _ in range(args.steps * reps):
            step()
        barrier(); el_s = time.perf_counter() - ts0
        if world > 1:
            te = torch.tensor([el_s], dtype=torch.float64, device=dev); dist.all_reduce(te, op=dist.ReduceOp.MAX); el_s = float(te.item())
        kts = st.get_timing(dev); st.enable_timing(False)
        sustained = {"value": H * W * args.steps * reps / el_s, "unit": "rays/s", "steps": args.steps * reps, "seconds": el_s,
                     "ms_per_step": 1e3 * el_s / (args.steps * reps),
                     "phase_ms": {k_: kts[k_][0] / max(kts[k_][1], 1) for k_ in ("build", "fwd", "bwd")}}
    steps_long = args.steps * max(reps, 1)                            # length of the other secondary windows (drop-in path, varying frame)
    # ---- the timed window: EXACTLY --steps steps, barrier + synchronize on both sides, max over ranks.
    # The library's region timers (HIP events on the launch stream around build / forward / colour pass / backward) run INSIDE it, on every
    # --time-every-th step: an event record between two kernels costs ~5 us of pipeline, eight per step were 3-5 % of the step they measure
    steps_run = args.steps
    st.enable_timing(not args.graph)                                  # (they record events between kernels: not inside a graph)
    tr.enable_phase_timing(True, every=max(1, args.time_every))
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps_run):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    kt = st.get_timing(dev)
    st.enable_timing(False)
    phases = tr.phase_timing()
    tr.enable_phase_timing(False)
    if sustained is not None and any(kt[k_][1] == 0 for k_ in ("build", "fwd", "bwd")):
        kt = kts                                                      # a window shorter than --time-every steps holds no sample: the sustained window's averages

    # ---- N > 1: every rank's own compute (its library regions: build + forward + backward), and the unsharded step of the same workload on rank 0
    per_rank_ms, value_n1 = None, None
    if world > 1:
        mine = torch.tensor([kt[k_][0] / max(kt[k_][1], 1) for k_ in ("build", "fwd", "bwd")], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_ms = [{"build": float(a_[0]), "forward": float(a_[1]), "backward": float(a_[2]), "sum": float(a_.sum())} for a_ in allr]
        if rank == 0:
            tr1 = ShardedTracer(exchange=args.exchange, world=1, rank=0)
            for kv in args.opt:
                k_, v_ = kv.split("="); tr1.backend.state.set_option(k_, int(v_))
            tr1.backend.state.refit_interval = max(args.refit_every, 0)      # the same step as the sharded one
            def step1():
                tr1.forward(ray_o, ray_d, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], deg, bg, rebuild=not args.no_build_in_step, cull_key="bench-frame")
                tr1.backward(t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], deg, bg, dL)
            if args.no_build_in_step:
                tr1.backend.build(t["means"], t["scales"], t["rotations"], t["opacities"])
            for _ in range(max(args.warmup, 3) + 8):
                step1()
            n1_steps = max(args.steps, steps_run)                     # the length of the headline's window
            torch.cuda.synchronize(); t1_ = time.perf_counter()
            for _ in range(n1_steps):
                step1()
            torch.cuda.synchronize(); e1_ = time.perf_counter() - t1_
            value_n1 = {"value": H * W * n1_steps / e1_, "unit": "rays/s", "steps": n1_steps, "ms_per_step": 1e3 * e1_ / n1_steps,
                        "note": "INDICATIVE: the unsharded step of the same workload on rank 0's device, timed right behind the sharded window while the other ranks "
                                "wait in a barrier; the driver's own N=1 run is the reference for scaling"}
            del tr1
        barrier()

    other = None
    if args.both_paths and world == 1:
        ostep = step_direct if args.via == "tracer" else step_tracer
        for _ in range(max(args.warmup, 1)):
            ostep()
        barrier(); t1 = time.perf_counter()
        for _ in range(steps_long):
            ostep()
        barrier(); other = H * W * steps_long / (time.perf_counter() - t1)

    # ---------------- the training configuration of the step: deferred hit weights (Tracer(deferred_accum=True) / renderer.deferred_accum):
    # the forward issues no float atomic per composited hit, the backward's Gaussian-ordered reduction writes the same sums (train.py reads
    # them after loss.backward() only: :156, :219).  NOT the headline: `value` keeps the reference's contract (weights complete at the forward)
    deferred = None
    if world == 1 and not args.no_deferred:
        tr_d = ShardedTracer(exchange=args.exchange, deferred_accum=True)
        for kv in args.opt:
            k_, v_ = kv.split("="); tr_d.backend.state.set_option(k_, int(v_))
        tr_d.backend.state.set_option("timing_every", max(1, args.time_every))

        def step_deferred():
            out_, _ = tr_d.forward(ray_o, ray_d, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], deg, bg, rebuild=not args.no_build_in_step, cull_key="bench-frame")
            return out_, tr_d.backward(t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], deg, bg, dL)
        if args.no_build_in_step:
            tr_d.backend.build(t["means"], t["scales"], t["rotations"], t["opacities"])
        for _ in range(max(args.warmup, 3)):
            step_deferred()
        tr_d.backend.state.enable_timing(True)
        barrier(); td = time.perf_counter()
        for _ in range(steps_long):
            out_d, g_d = step_deferred()
        barrier(); el_d = time.perf_counter() - td
        ktd = tr_d.backend.state.get_timing(dev); tr_d.backend.state.enable_timing(False)
        tr_d.backend.state.check(dev, wait=True)
        out_x, g_x = step()
        torch.cuda.synchronize()
        acc_err = float((g_d["accum"].double() - g_x["accum"].double()).norm() / g_x["accum"].double().norm().clamp_min(1e-300))
        deferred = {"value": H * W * steps_long / el_d, "unit": "rays/s", "steps": steps_long, "ms_per_step": 1e3 * el_d / steps_long,
                    "phase_ms": {k_: ktd[k_][0] / max(ktd[k_][1], 1) for k_ in ("build", "fwd", "bwd")},
                                        "accum_rel_l2_vs_forward_atomics": acc_err,
                    # (two tracer states: the image is bit-identical for equal slab partitions -- tools/check_identical.py -- and equal to the rounding of
                    # the per-slab partial sums otherwise: the learned first-slab widths of the two states differ)
                    "image_max_abs_diff": float((out_d - out_x).abs().max()),
                    "note": "library option deferred_accum=1 (lrt_backward_accum): no accum atomics in the forward, k_bwd_reduce4 writes the column"}
        del tr_d

    # ---------------- the bit-reproducible step (ShardedTracer(deterministic=True) / train --deterministic): gradient sums in a fixed order (a second pass over the
    # records in k_bk_sort, k_bwd_fixup), a forward without learnt tables that waits for its status words.  NOT the headline: what reproducibility costs.
    determ = None
    if world == 1 and not args.no_deferred and not args.no_build_in_step and args.refit_every <= 0:
        tr_t = ShardedTracer(exchange=args.exchange, deterministic=True)
        tr_t.backend.state.set_option("timing_every", max(1, args.time_every))

        def step_det():
            out_, _ = tr_t.forward(ray_o, ray_d, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], deg, bg, cull_key="bench-frame")
            return out_, tr_t.backward(t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], deg, bg, dL)
        for _ in range(max(args.warmup, 3)):
            out_t0, g_t0 = step_det()
        g_t0 = {k_: v_.clone() for k_, v_ in g_t0.items()}; out_t0 = out_t0.clone()
        n_t = max(steps_long // 3, 50)
        tr_t.backend.state.enable_timing(True)
        barrier(); tt = time.perf_counter()
        for _ in range(n_t):
            out_t, g_t = step_det()
        barrier(); el_t = time.perf_counter() - tt
        ktt = tr_t.backend.state.get_timing(dev); tr_t.backend.state.enable_timing(False)
        determ = {"value": H * W * n_t / el_t, "unit": "rays/s", "steps": n_t, "ms_per_step": 1e3 * el_t / n_t,
                  "phase_ms": {k_: ktt[k_][0] / max(ktt[k_][1], 1) for k_ in ("build", "fwd", "bwd")},
                  "bit_identical_over_the_window": bool(torch.equal(out_t, out_t0) and all(torch.equal(g_t[k_], g_t0[k_]) for k_ in g_t0)),
                  "note": "library option deterministic=1 (+ deferred_accum): runs ordered by ray, pieces of long runs added in wave order, no learnt first-slab widths / carried order / lagged box, one host wait per forward"}
        del tr_t

    # ---------------- the step with a full Morton sort in EVERY build (library option carry_order=0; the headline's builds keep the order of the last
    # sort for up to 32 builds -- what a training loop's builds do between optimizer steps -- and sort again when it has decayed)
    full_sort = None
    if world == 1 and not args.no_deferred and not args.no_build_in_step and args.refit_every <= 0:
        st.set_option("carry_order", 0)
        for _ in range(3):
            step()
        st.enable_timing(True); st.get_timing(dev)
        barrier(); tf = time.perf_counter()
        for _ in range(max(steps_long // 3, 50)):
            step()
        barrier(); el_f = time.perf_counter() - tf
        ktf = st.get_timing(dev); st.enable_timing(False)
        st.set_option("carry_order", 1)
        for _ in range(3):
            step()
        full_sort = {"value": H * W * max(steps_long // 3, 50) / el_f, "unit": "rays/s", "steps": max(steps_long // 3, 50), "ms_per_step": 1e3 * el_f / max(steps_long // 3, 50),
                     "phase_ms": {k_: ktf[k_][0] / max(ktf[k_][1], 1) for k_ in ("build", "fwd", "bwd")}, "note": "library option carry_order=0: every build sorts (k_morton + 3 onesweep passes + k_make_tree)"}

    # ---------------- the same step on a frame that changes every iteration (--vary)
    varying = None
    if args.vary and world == 1:
        n_pose = 4
        # the sensor stays inside the scene's empty cylinder (r < 2 m around the origin): a pose inside the clutter would make hundreds of
        # "near rays" (a quad closer than 0.2 m: the literal K-buffer replay of lrt_near.inc), which is a property of that pose, not of a moving frame
        yaws = [0.0, 0.35, -0.6, 1.3]; shifts = [(0.0, 0.0, 0.0), (0.30, -0.20, 0.05), (-0.35, 0.25, -0.03), (0.15, 0.40, 0.02)]
        if args.pose_inside:
            # --pose-inside: the sensor 0.1 m from a clutter Gaussian's centre, along its normal (train.py's real sensors sit inside geometry): the rays
            # that start closer than 0.2 m to a quad are the literal K-buffer replay's (k_fwd_near) -- the case the default poses avoid
            m_ = sc["means"]; r_ = np.hypot(m_[:, 0], m_[:, 1])
            cand = np.nonzero((r_ > 5.0) & (r_ < 40.0) & (m_[:, 2] > 0.0) & (m_[:, 2] < 2.0))[0]
            pick = cand[np.linspace(0, len(cand) - 1, n_pose).astype(int)]
            q_ = sc["rotations"][pick]; q_ = q_ / np.linalg.norm(q_, axis=1, keepdims=True)
            nrm = np.stack([2 * (q_[:, 1] * q_[:, 3] + q_[:, 0] * q_[:, 2]), 2 * (q_[:, 2] * q_[:, 3] - q_[:, 0] * q_[:, 1]),
                            1 - 2 * (q_[:, 1] ** 2 + q_[:, 2] ** 2)], 1)
            shifts = [tuple(float(x) for x in (m_[i_] + 0.1 * n_)) for i_, n_ in zip(pick, nrm)]
        poses = []
        for yw, sh in zip(yaws, shifts):
            c_, s_ = float(np.cos(yw)), float(np.sin(yw))
            R = torch.tensor([[c_, -s_, 0.0], [s_, c_, 0.0], [0.0, 0.0, 1.0]], device=dev)
            poses.append(((ray_o + torch.tensor(sh, device=dev)).contiguous(), (ray_d @ R.T).contiguous()))
        gen = torch.Generator(device=dev).manual_seed(7)
        # Adam-sized steps (position lr 1.6e-4 x scene extent ~ 1e-2 m early in training, 1e-3 later; scaling lr 5e-3 on the log; opacity lr 5e-2 on the logit)
        d_means = [1e-3 * torch.randn(t["means"].shape, device=dev, generator=gen) for _ in range(2)]
        d_scale = [1.0 + 5e-3 * torch.randn(t["scales"].shape, device=dev, generator=gen) for _ in range(2)]
        d_opac = [0.01 * torch.randn(t["opacities"].shape, device=dev, generator=gen) for _ in range(2)]
        saved = {k: t[k].clone() for k in ("means", "scales", "opacities")}

        def step_vary(i):
            k_ = i & 1; sgn = 1.0 if (i >> 1) & 1 == 0 else -1.0                       # +a +b -a -b: the parameters stay where they were on average
            t["means"].add_(d_means[k_], alpha=sgn)
            t["scales"].mul_(d_scale[k_] if sgn > 0 else 1.0 / d_scale[k_])
            t["opacities"].add_(d_opac[k_], alpha=sgn).clamp_(0.01, 0.99)
            o_, d_ = poses[i % n_pose]
            out_, _ = tr.forward(o_, d_, t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], deg, bg, rebuild=not args.no_build_in_step,
                                 cull_key=("bench-pose", i % n_pose))
            tr.backward(t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], deg, bg, dL)

        for i in range(8):
            step_vary(i)
        st.enable_timing(True)
        barrier(); tv = time.perf_counter()
        for i in range(steps_long):
            step_vary(i)
        barrier(); el_v = time.perf_counter() - tv
        ktv = st.get_timing(dev); st.enable_timing(False)
        st.check(dev, wait=True)                                                      # an overflow of a speculated capacity would surface here
        near = None
        if args.pose_inside:
            near = []
            for i in range(n_pose):
                step_vary(i); torch.cuda.synchronize()
                near.append(int(st.get_option("near_rays_last", dev)))
        for k, v in saved.items():
            t[k].copy_(v)
        varying = {"value": H * W * steps_long / el_v, "unit": "rays/s", "steps": steps_long, "ms_per_step": 1e3 * el_v / steps_long,
                   "poses": n_pose, "yaw_rad": yaws, "shift_m": shifts, "pose_inside": bool(args.pose_inside), "near_rays_per_frame": near,
                   "phase_ms": {k_: ktv[k_][0] / max(ktv[k_][1], 1) for k_ in ("build", "fwd", "bwd")},
                   "parameter_step": "means += N(0, 1e-3 m), scales *= 1 + N(0, 5e-3), opacities += N(0, 0.01) per step (signs alternate); the three "
                                     "in-place updates (~10 us of elementwise kernels) are inside the timed window"}
        step()                                                                         # back on the fixed frame for the statistics step below

    # ---------------- one instrumented step for the traversal statistics (untimed)
    if not args.no_stats_step:
        st.enable_stats(True)
    out, g = step()
    torch.cuda.synchronize()
    hs = st.get_stats(dev) if not args.no_stats_step else {}
    st.enable_stats(False)

    cks = None
    if args.check_sum:
        # owner exchange: a Gaussian's gradient is complete on its owner only -> sum over the owned rows, then over the ranks
        def _abs_sum(x, rows=None):
            x = x.double().abs().reshape(x.shape[0], -1)
            return (x[rows] if rows is not None else x).sum()
        own = (tr.last_owner == rank) if (world > 1 and tr.last_exchange == "owner") else None
        vals = torch.stack([out.double().abs().sum(), _abs_sum(g["means"], own), _abs_sum(g["shs"], own),
                            (g["accum"].double()[own] if own is not None else g["accum"].double()).sum()])
        if own is not None:
            part = vals[1:].clone(); dist.all_reduce(part, op=dist.ReduceOp.SUM); vals[1:] = part
        cks = dict(zip(("out", "d_means", "d_shs", "accum"), [float(v) for v in vals.tolist()]))
    if rank == 0:
        n_rays = H * W
        ms_per_step = 1e3 * elapsed / steps_run
        value = n_rays * steps_run / elapsed
        # ---- roofline of the dominant kernel (per launch, this rank's slab)
        a, b = getattr(tr, "_slab", None) or column_slab(W, rank, world)      # --via tracer never runs ShardedTracer.forward
        rays_local = H * (b - a)
        stats_path = os.path.join(REPO, "tests", "golden", "s1m_stats.json")
        if args.workload == "s1m" and os.path.exists(stats_path):
            gs = json.load(open(stats_path))
            C, K = gs["C_mean_candidates_per_ray"], gs["K_mean_composited_per_ray"]
            ck_src = "oracle (tests/golden/s1m_stats.json)"
        else:
            # the bucketed backward replays the hit record: nothing is counted twice (S1M: composited = 3,939,575 = 30.06 x 131,072 exactly)
            C = hs.get("candidates", 0) / max(rays_local, 1); K = hs.get("composited", 0) / max(rays_local, 1)
            ck_src = "HIP counters of this run (forward traversal of the instrumented step)"
        bf, bb = algorithmic_bytes(C, K, deg)
        ms_f = kt["fwd"][0] / max(kt["fwd"][1], 1); ms_b = kt["bwd"][0] / max(kt["bwd"][1], 1)
        ms_build = kt["build"][0] / max(kt["build"][1], 1)
        dom = "backward (k_bk_count .. k_bwd_prep .. k_bk_sort + k_bwd_reduce4)" if ms_b >= ms_f else "forward (k_fwd_cr4 + k_fwd_colour)"
        dom_ms = max(ms_b, ms_f); dom_bytes = (bb if ms_b >= ms_f else bf) * rays_local
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        ms_c = kt["colour"][0] / max(kt["colour"][1], 1) if "colour" in kt else 0.0
        n_sh = 12 * (deg + 1) ** 2
        # algorithmic bytes per ray of the SURVEY 8(d) model, split by kernel: the trace kernel reads the rays, the candidates' splat
        # parameters and adds the hit weights (8 B read-modify-write per composited hit); the colour pass reads the SH rows; the build
        # is not part of the SURVEY model (its node bytes are "implementation specific"): inputs + records + nodes of the LBVH, per step
        P_ = int(sc["means"].shape[0])
        alg = {"k_fwd_cr4": (24 + 36 + 40 * C + 8 * K) * rays_local, "k_fwd_colour": n_sh * K * rays_local,
               "backward": bb * rays_local, "build": float(P_) * (40 + 64) + (P_ / 8.0 / 7.0) * 256 * 2}
        live_ms = {"k_fwd_cr4": max(ms_f - ms_c, 0.0), "k_fwd_colour": ms_c, "backward": ms_b, "build": ms_build}
        traffic = None; traffic_raw = None; valu = None; step_traffic = None; step_traffic_raw = None; traffic_note = None
        per_kernel = {k: {"algorithmic_bytes": alg[k], "live_ms": live_ms[k],
                          "frac_algorithmic": (alg[k] / (live_ms[k] * 1e-3) / HBM_PEAK_BYTES_PER_S) if live_ms[k] > 0 else None} for k in alg}
        per_kernel["k_fwd_cr4"]["live_ms_note"] = "forward region minus the colour pass (includes k_fwd_near, a few us)"
        # the SURVEY 8(d) backward term prices 58 read-modify-write atomics per hit (backward.cu:615,659-669) that the replay design never
        # performs (one row store per touched Gaussian): its "fraction" is a model figure, NOT traffic -- the counter fraction is beside it
        per_kernel["backward"]["model_fraction_not_traffic"] = per_kernel["backward"].pop("frac_algorithmic")
        tp = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(tp):
            try:
                from lidar_rt_amd.build import source_hash
                pj = json.load(open(tp))
                if pj.get("_meta", {}).get("csrc_sha") == source_hash() and "regions" in pj and args.workload == "s1m" and world == 1:
                    regs, kern = pj["regions"], pj["kernels"]
                    region_key = "backward" if ms_b >= ms_f else "forward"
                    traffic = regs[region_key]["corrected_bytes_per_step"]; traffic_raw = regs[region_key]["raw_bytes_per_step"]
                    step_traffic = pj["step"]["corrected_bytes_per_step"]; step_traffic_raw = pj["step"]["raw_bytes_per_step"]
                    cr4 = [k for k in kern if k.startswith("k_fwd_cr4") and kern[k]["region"] == "forward"]
                    valu = max((kern[k].get("valu_issue_frac", 0.0) for k in cr4), default=None)
                    src_rows = {"k_fwd_cr4": cr4, "k_fwd_colour": [k for k in kern if k.startswith("k_fwd_colour")], "backward": regs["backward"]["kernels"], "build": regs["build"]["kernels"]}
                    for name, ks in src_rows.items():
                        ks = [k for k in ks if k in kern and kern[k]["launches_per_step"]]
                        corr = sum(kern[k]["corrected_bytes_per_launch"] * kern[k]["launches_per_step"] for k in ks)
                        raw = sum(kern[k]["raw_bytes_per_launch"] * kern[k]["launches_per_step"] for k in ks)
                        us = sum((kern[k]["avg_us"] or 0.0) * kern[k]["launches_per_step"] for k in ks)
                        per_kernel[name].update({"counter_bytes": corr, "counter_bytes_uncorrected": raw, "profile_us": us,
                                                 "frac_counters": corr / (us * 1e-6) / HBM_PEAK_BYTES_PER_S if us > 0 else None,
                                                 "frac_counters_uncorrected": raw / (us * 1e-6) / HBM_PEAK_BYTES_PER_S if us > 0 else None,
                                                 "profile_kernels": ks})
                    traffic_note = (f"rocprofv3 PMC passes of profile '{pj['_meta'].get('tag')}' (profiles/pmc_traffic.json, profiles/{pj['_meta'].get('tag')}_summary.md), "
                                    f"same kernel sources (hash {source_hash()}); per step = per-launch average x launches per step")
                elif pj.get("_meta", {}).get("csrc_sha") != source_hash():
                    traffic_note = "profiles/pmc_traffic.json was collected for other kernel sources (hash mismatch): not reported"
                else:
                    traffic_note = "profiles/pmc_traffic.json was collected on S1M with one GPU: not reported for this configuration"
            except Exception as ex:
                traffic_note = f"profiles/pmc_traffic.json unreadable: {ex}"
        roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_BYTES_PER_S / 1e9, "unit": "GB/s",
                "frac": achieved / (HBM_PEAK_BYTES_PER_S / 1e9), "traffic": traffic,
                # the companion of `frac`: counter-measured HBM bytes of the same region (committed rocprofv3 PMC profile of the same sources) over
                # this run's live time of that region
                "frac_counters": (traffic / (dom_ms * 1e-3) / HBM_PEAK_BYTES_PER_S) if (traffic and dom_ms > 0) else None,
                # `traffic` = counter bytes per step of the dominant region's kernels with the guide's gfx950 correction (2 x FETCH_SIZE +
                # WRITE_SIZE); the uncorrected sum beside it; which applies to which access pattern: profiles/r04_fetch_calibration.md
                "traffic_uncorrected": traffic_raw,
                # supplementary (committed rocprofv3 PMC profile): the trace kernel is bound by VALU issue, not by HBM
                "valu_issue_frac": valu,
                "algorithmic_bytes_per_ray": {"fwd": bf, "bwd": bb, "C": C, "K": K, "source": ck_src},
                "avg_kernel_ms": {"build_region": ms_build, "trace_fwd": ms_f, "trace_bwd": ms_b, "colour_pass": ms_c},
                "avg_kernel_ms_note": f"HIP events on the launch stream inside the timed window, recorded on every {max(1, args.time_every)}. step of it ({kt['fwd'][1]} forward launches timed)",
                # per kernel / region: algorithmic bytes (SURVEY 8(d) split), live HIP-event time of this run, and -- from the committed
                # profile of the same sources -- counter bytes (corrected / raw), rocprofv3 kernel time and the fractions of 8 TB/s
                "per_kernel": per_kernel,
                "traffic_source": traffic_note,
                # whole step: the SURVEY 8(d) byte model (its backward term, 58 read-modify-write atomics per hit, is traffic the
                # replay design does not move), and beside it the counter-measured bytes of ALL kernels of a step
                "whole_step_frac": (bf + bb) * n_rays / (ms_per_step * 1e-3) / HBM_PEAK_BYTES_PER_S,
                "whole_step_frac_counters": (step_traffic / (ms_per_step * 1e-3) / HBM_PEAK_BYTES_PER_S) if (step_traffic and world == 1) else None,
                "whole_step_traffic": step_traffic if world == 1 else None,
                "whole_step_frac_counters_uncorrected": (step_traffic_raw / (ms_per_step * 1e-3) / HBM_PEAK_BYTES_PER_S) if (step_traffic_raw and world == 1) else None,
                "whole_step_traffic_uncorrected": step_traffic_raw if world == 1 else None}
        res = {
            "metric": "LiDAR rays/s fwd+bwd @1M Gaussians, 2048x64 sweep; % HBM roofline",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": steps_run, "steps_requested": args.steps, "warmup": args.warmup,
            "timed_window_s": elapsed,
            # the same step over a window of --min-seconds, run in front of the timed one
            "sustained": sustained,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl, "gaussians": int(sc["means"].shape[0]), "rays": [H, W], "sh_degree": deg,
                       "step": (("LBVH build (records and tree from the current parameters on the carried Morton order; a full sort every <= 32 builds or when the order has decayed) + " if args.refit_every <= 0 else f"LBVH refit ({args.refit_every} between rebuilds) + ") if not args.no_build_in_step else "") + "forward + backward"
                               + (f" + slab all_gather + gradient exchange '{tr.last_exchange}' (device-side: one pack launch, one all_gather, one apply launch; capacity overflows are flagged on the device and raised by the next step)" if world > 1 else ""),
                       "parallelism": f"azimuth-sector x{world}", "options": args.opt, "dist_backend": backend if world > 1 else None,
                       "gradient_exchange": tr.last_exchange, "via": args.via, "binding": _binding.BACKEND,
                       "slab_edges": (list(tr._edges) if getattr(tr, "_edges", None) is not None else ([column_slab(W, r, world)[0] for r in range(world)] + [W] if world > 1 else None)),
                       "hip_graph": ({"replays": st.get_option("graph_hits", dev), "instantiated": st.get_option("graph_captures", dev)} if args.graph else None)},
            "roofline": roof,
            # per-phase GPU time per step on rank 0 (HIP events): LBVH build / forward trace / backward; N > 1: + slab all_gather and
            # gradient exchange (torch events around the collectives and their pack / unpack kernels)
            "phase_ms": {"build": ms_build, "forward": ms_f, "backward": ms_b, **{k: v for k, v in phases.items()}},
            "hip_counters_per_step": {k: v for k, v in hs.items()},
        }
        if world > 1:
            res["value_n1"] = value_n1
            res["per_rank_compute_ms"] = per_rank_ms
            res["exchange_ms"] = {"slab_all_gather": phases.get("slab_all_gather"), "gradient_exchange": phases.get("gradient_exchange")}
            res["speedup_vs_n1_indicative"] = (value / value_n1["value"]) if (value_n1 and not args.graph and args.via == "direct") else None
        if other is not None:
            res["drop_in_path" if args.via == "direct" else "direct_path"] = {"value": other, "unit": "rays/s", "steps": steps_long}
        if varying is not None:
            res["value_varying"] = varying
        if deferred is not None:
            res["value_deferred_accum"] = deferred
        if determ is not None:
            res["value_deterministic"] = determ
        if full_sort is not None:
            res["value_sort_every_build"] = full_sort
        if args.check_sum:
            res["checksums"] = cks
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(sc, ro, rd, deg, bg_np, dL_np)
            except Exception as ex:  # the oracle is test infrastructure; never fail the bench on it
                res["cpu_baseline"] = {"value": None, "unit": "rays/s", "cores": 0, "kind": "port", "sample": f"failed: {ex}"}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
