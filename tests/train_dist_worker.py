"""Worker for the sharded TRAINING-STEP tests (tests/test_distributed_cpu.py on CPU with the oracle-backed stand-in and gloo;
tests/test_sharded_gpu.py on one GPU with the HIP backend and gloo): K `training.training_step`s incl. one densification through
`renderer.sharded`, parameters of every rank written to <out>.rank<r>.npz.  Without RANK in the environment it runs single-process."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from lidar_rt_amd import scenes, training, renderer          # noqa: E402
from lidar_rt_amd.parallel import ShardedTracer               # noqa: E402


def main():
    out_path, device, exchange = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "sparse")
    multi = "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1
    if multi:
        dist.init_process_group(backend="gloo")
    rank = dist.get_rank() if multi else 0
    dev = torch.device(device)
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
        backend = None                                         # the product's HipBackend
    else:
        from tests.oracle_backend import OracleBackend
        backend = OracleBackend()
        renderer.use_fused_preprocess = False                  # the fused pre-processing is a HIP operator
    P, H, W = (1500, 6, 45) if dev.type == "cpu" else (20000, 16, 190)
    sc = scenes.make_scene(P, seed=4, radius_scale=0.2 if dev.type == "cpu" else 0.35)
    t = lambda a: torch.as_tensor(a, device=dev)
    asset = training.GaussianAsset.from_tensors(t(sc["means"]), t(sc["shs"][:, :1]).contiguous(), t(sc["shs"][:, 1:]).contiguous(),
                                                torch.log(t(sc["scales"])), t(sc["rotations"]), training.inverse_sigmoid(t(sc["opacities"])),
                                                max_sh_degree=3, extent=8.0)
    asset.active_sh_degree = 3
    scene = training.GaussianScene([asset])
    opt = training.default_options()
    opt.lambda_cd = 0.0 if dev.type == "cpu" else opt.lambda_cd    # the Chamfer operator is HIP only
    opt.densify_from_iter, opt.densification_interval, opt.densify_until_iter = 1, 3, 100
    opt.densify_grad_threshold = float(os.environ.get("LRT_TEST_GRAD_THRESHOLD", "2e-6"))
    opt.densify_scale_threshold, opt.thresh_opa_prune = 0.0125, 0.03      # both clones and splits, and some opacity pruning
    scene.training_setup(opt)
    frames = training.RangeFrames()
    rng = np.random.default_rng(7)
    for f in range(3):
        o, d = scenes.range_rays(H, W, (np.radians(-24.9), np.radians(2.0)), scenes.pose_matrix((0.1 * f, 0.05 * f, 0.0), yaw=0.03 * f), "KITTI")
        depth = (4.0 + 2.0 * np.sin(np.linspace(0, 6, W))[None, :] + 0.3 * rng.standard_normal((H, W))).astype(np.float32)
        inten = np.clip(0.5 + 0.2 * rng.standard_normal((H, W)), 0, 1).astype(np.float32)
        mask = rng.uniform(size=(H, W)) < 0.8
        frames.add_frame(f, t(o), t(d), t(depth), t(inten), t(mask))
    bg = t(scenes.BG_DEFAULT)
    tr = ShardedTracer(backend=backend, exchange=exchange)
    if os.environ.get("LRT_TEST_FIRST_CAP"):
        tr.first_cap = int(os.environ["LRT_TEST_FIRST_CAP"])     # too small on purpose: the in-step verification must re-run the exchange
    renderer.sharded = tr
    torch.manual_seed(1234)                                    # the split of densify_and_prune samples: the same draws on every rank
    log = []
    # LRT_TEST_BAD_CULL = "once:<it>": the first culled build of iteration <it> is sized for 64 primitives -- far too small, it loses primitives
    # (error bit 8): training_step must notice BEFORE the optimizer step and redo the step;
    # "always:<it>": every build of that iteration is under-sized: training_step must raise with the parameters untouched
    bad_mode, bad_it = (os.environ.get("LRT_TEST_BAD_CULL", ":0").split(":") + ["0"])[:2]
    redone, raised_clean = 0, -1
    for it in range(1, 6):
        if bad_mode == "once" and it == int(bad_it):
            sizing_ = tr._cull_sizing
            def too_small_once(key, _o=sizing_):
                _o(key); tr.backend.state.set_option("cull_next", 64); tr._cull_sizing = _o
            tr._cull_sizing = too_small_once
        if bad_mode == "always" and it == int(bad_it):
            orig_sizing = tr._cull_sizing
            def too_small(key, _o=orig_sizing):
                _o(key); tr.backend.state.set_option("cull_next", 64)
            tr._cull_sizing = too_small
            before = {n: p.detach().clone() for n, p in asset._params().items()}
            m_before = asset.optimizer.state[asset._xyz]["exp_avg"].clone() if asset._xyz in asset.optimizer.state else None
            try:
                training.training_step(scene, frames, it % 3, it, opt, bg)
                raised_clean = 0
            except Exception as ex:                                     # LrtError on every rank alike
                same = all(torch.equal(before[n], p.detach()) for n, p in asset._params().items())
                if m_before is not None:
                    same = same and torch.equal(m_before, asset.optimizer.state[asset._xyz]["exp_avg"])
                raised_clean = 1 if (same and "NOT taken" in str(ex)) else 0
            tr._cull_sizing = orig_sizing
            for g_ in scene.gaussians_assets:
                for p_ in g_._params().values():
                    p_.grad = None
        r = training.training_step(scene, frames, it % 3, it, opt, bg)
        redone += int(r.get("step_redone", 0))
        log.append([float(r["loss"]), float(r["points"])] + [float(x) for x in r["densify"]])
    tr.check()
    pr = {n: p.detach().cpu().numpy() for n, p in asset._params().items()}
    st = asset.optimizer.state[asset._xyz]
    np.savez(out_path + f".rank{rank}.npz", log=np.asarray(log), m_xyz=st["exp_avg"].cpu().numpy(), v_xyz=st["exp_avg_sq"].cpu().numpy(),
             reruns=np.asarray([tr.exchange_reruns]), redone=np.asarray([redone]), raised_clean=np.asarray([raised_clean]), **pr)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
