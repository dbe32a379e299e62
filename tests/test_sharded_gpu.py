"""GPU: the N>1 code path end to end on ONE device -- two ranks share cuda:0 and use gloo (RCCL needs one GPU per
rank); the sharded result must equal the single-rank result.  The real multi-GPU run is the driver's."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(world, **extra_env):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", LRT_SINGLE_DEVICE="1", LRT_DIST_BACKEND="gloo", **extra_env)
    if world == 1:
        cmd = [sys.executable, os.path.join(REPO, "bench.py")]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", "29611", os.path.join(REPO, "bench.py")]
    cmd += ["--gpus", str(world), "--steps", "2", "--warmup", "1", "--workload", "s10k", "--no-cpu-baseline", "--check-sum", "--min-seconds", "0", "--no-vary", "--no-both-paths"]
    out = subprocess.run(cmd, check=True, env=env, cwd=REPO, timeout=600, capture_output=True, text=True).stdout
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


def test_two_ranks_on_one_gpu_match_single_rank():
    a, b = _bench(1), _bench(2)
    assert b["n_gpus"] == 2 and b["scaling"] == "strong"
    assert b["config"]["gradient_exchange"] == "sparse"               # the default: replicated result (HIP list / pack / zero / add kernels)
    for k in ("out", "d_means", "d_shs", "accum"):
        assert abs(a["checksums"][k] - b["checksums"][k]) <= 2e-5 * max(abs(a["checksums"][k]), 1e-12), (k, a["checksums"], b["checksums"])


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run (the form the driver uses at N = 1) re-executes itself under the launcher:
    the line says two ranks ran and carries the N = 1 value of the same workload, every rank's compute and the exchange times."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", LRT_SINGLE_DEVICE="1", LRT_DIST_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--workload", "s10k", "--no-cpu-baseline", "--min-seconds", "0"]
    out = subprocess.run(cmd, check=True, env=env, cwd=REPO, timeout=600, capture_output=True, text=True).stdout
    b = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert b["n_gpus"] == 2 and b["steps"] == 4 and b["steps_requested"] == 4
    assert b["value_n1"]["value"] > 0 and len(b["per_rank_compute_ms"]) == 2 and all(r["sum"] > 0 for r in b["per_rank_compute_ms"])
    assert set(b["exchange_ms"]) == {"slab_all_gather", "gradient_exchange"} and b["speedup_vs_n1_indicative"] > 0


def test_rccl_world_of_two_when_two_devices_are_visible():
    """The first real multi-GPU step: two ranks, one device each, backend "nccl" (RCCL over xGMI), through bench.py's own launcher.  Skipped
    (with the reason) on a one-GPU box -- the 8-GPU scaling run is the driver's."""
    if torch.cuda.device_count() < 2:
        pytest.skip(f"{torch.cuda.device_count()} visible HIP device(s): RCCL with two ranks needs two")
    a = _bench(1)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LRT_SINGLE_DEVICE", "LRT_DIST_BACKEND")}
    env.update(MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--workload", "s10k", "--no-cpu-baseline", "--check-sum", "--min-seconds", "0"]
    out = subprocess.run(cmd, check=True, env=env, cwd=REPO, timeout=600, capture_output=True, text=True).stdout
    b = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert b["n_gpus"] == 2 and b["config"]["dist_backend"] == "nccl"
    for k in ("out", "d_means", "d_shs", "accum"):
        assert abs(a["checksums"][k] - b["checksums"][k]) <= 2e-5 * max(abs(a["checksums"][k]), 1e-12), (k, a["checksums"], b["checksums"])


def test_rebalanced_slabs_give_the_single_rank_results():
    """The slab edges move towards equal compute time per rank (every LRT_BALANCE_EVERY steps, from the times every rank sends along with
    its slab): whatever split the ranks agree on, image and gradients equal the single-rank run's."""
    a = _bench(1)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", LRT_SINGLE_DEVICE="1", LRT_DIST_BACKEND="gloo", LRT_BALANCE_EVERY="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1", "--master-port", "29615",
           os.path.join(REPO, "bench.py"), "--gpus", "3", "--steps", "9", "--warmup", "2", "--workload", "s10k", "--no-cpu-baseline", "--check-sum", "--min-seconds", "0"]
    out = subprocess.run(cmd, check=True, env=env, cwd=REPO, timeout=600, capture_output=True, text=True).stdout
    b = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert b["config"].get("slab_edges") is not None                  # the split that was in use at the end (may or may not have moved)
    for k in ("out", "d_means", "d_shs", "accum"):
        assert abs(a["checksums"][k] - b["checksums"][k]) <= 2e-5 * max(abs(a["checksums"][k]), 1e-12), (k, a["checksums"], b["checksums"])


def test_three_ranks_with_ray_culled_builds_match_single_rank():
    """The default from eight ranks on (every rank builds the LBVH for its own slab's rays, sized speculatively from the previous frame)
    end to end through ShardedTracer, forced on at three ranks."""
    a, b = _bench(1), _bench(3, LRT_CULL_BUILD="1")
    assert b["n_gpus"] == 3
    for k in ("out", "d_means", "d_shs", "accum"):
        assert abs(a["checksums"][k] - b["checksums"][k]) <= 2e-5 * max(abs(a["checksums"][k]), 1e-12), (k, a["checksums"], b["checksums"])


def test_replicated_exchanges_on_one_gpu_match_single_rank():
    a = _bench(1)
    for ex in ("owner", "dense"):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", LRT_SINGLE_DEVICE="1", LRT_DIST_BACKEND="gloo")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29613",
               os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "s10k", "--no-cpu-baseline", "--check-sum",
               "--min-seconds", "0", "--exchange", ex]
        out = subprocess.run(cmd, check=True, env=env, cwd=REPO, timeout=600, capture_output=True, text=True).stdout
        b = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        assert b["config"]["gradient_exchange"] == ex
        for k in ("out", "d_means", "d_shs", "accum"):
            assert abs(a["checksums"][k] - b["checksums"][k]) <= 2e-5 * max(abs(a["checksums"][k]), 1e-12), (ex, k)


def test_rccl_preflight_world_of_one():
    """The collectives of ShardedTracer on the backend they were written for: backend "nccl" (= RCCL) with ONE rank, every exchange
    mode, collective code paths forced on.  all_gather_into_tensor / all_reduce / all_to_all_single execute on the driver's box at least
    once; the results equal the plain single-rank ones."""
    worker = os.path.join(REPO, "tests", "nccl_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, worker], check=True, env=env, cwd=REPO, timeout=600, capture_output=True, text=True).stdout
    res = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert res["backend"] == "nccl" and set(res["exchanges"]) == {"owner", "dense", "sparse"}
    assert all(v < 1e-6 for v in res["rel_err"].values()), res
    # the whole sharded step (incl. the device-side gathering exchange) was enqueued behind 0.4 s of GPU work without the host waiting
    assert res["gpu_still_busy_after_enqueue"] and res["sharded_step_host_s"] < 0.1, res


def test_an_overflow_on_one_rank_is_raised_by_all_ranks():
    """A device-side overflow on ONE rank (its speculatively sized ray-culled build loses primitives) must not leave the other ranks
    blocked in a collective: the status words travel with the slabs, and every rank raises at its next call -- the same step on all
    ranks -- then carries on."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1", "--master-port", "29619",
           os.path.join(REPO, "tests", "overflow_worker.py")]
    out = subprocess.run(cmd, check=True, env=env, cwd=REPO, timeout=120, capture_output=True, text=True).stdout
    logs = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])["logs"]
    assert logs[0] == logs[1] == logs[2], logs                      # every rank saw the same sequence: nobody hung, nobody raised alone
    assert logs[0].count("raised") == 1 and logs[0][:3] == ["ok", "ok", "ok"] and logs[0][-1] == "clean", logs


def test_sharded_training_steps_on_one_gpu_match_the_single_rank_run(tmp_path):
    """`training_step` unchanged on two ranks (both on cuda:0, gloo) through `renderer.sharded` and the HIP backend: 5 iterations incl.
    one densification.  The replicas must be bit-identical to each other and hold the single-rank run's number of Gaussians and
    parameters (the two runs sum the same per-hit gradients in different orders: compared statistically)."""
    import numpy as np
    worker = os.path.join(REPO, "tests", "train_dist_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env1 = {k: v for k, v in env.items() if k not in ("RANK", "WORLD_SIZE")}
    subprocess.run([sys.executable, worker, str(tmp_path / "single"), "cuda:0", "sparse"], check=True, env=env1, cwd=REPO, timeout=600)
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29623",
                    worker, str(tmp_path / "w2"), "cuda:0", "sparse"], check=True, env=env, cwd=REPO, timeout=600)
    one = np.load(str(tmp_path / "single.rank0.npz")); r0 = np.load(str(tmp_path / "w2.rank0.npz")); r1 = np.load(str(tmp_path / "w2.rank1.npz"))
    assert one["log"][2, 2] + one["log"][2, 3] > 0, "the case must densify"
    for k in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "m_xyz", "v_xyz"):
        np.testing.assert_array_equal(r0[k], r1[k])                                              # replicas never diverge
    np.testing.assert_array_equal(r0["log"][:, 1:], one["log"][:, 1:])                            # same P, same clone / split / prune counts
    for k in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"):
        a, b = r0[k].astype(np.float64), one[k].astype(np.float64)
        scale = max(np.abs(b).max(), 1.0)
        far = np.abs(a - b) > 1e-5 * scale                                 # the handful of Adam-amplified elements (see same_run below) ...
        assert far.mean() < 2e-3, (k, float(np.abs(a - b).max()))
        assert np.linalg.norm((a - b)[~far]) / np.linalg.norm(b) < 1e-4, k        # ... and the tight bound on everything else (rows lost from the exchange move thousands of elements, not a handful)


def test_an_undersized_culled_build_never_reaches_the_optimizer(tmp_path):
    """a speculatively sized culled build that loses primitives used to be reported one step late -- after
    ``optimizer.step()`` had consumed the incomplete gradients.  ``training_step`` now reads the status words every rank gathered between
    ``loss.backward()`` and ``scene.optimize``: (a) a wrong size ONCE -> the step is redone with an exact build and the run ends with
    bit-identical parameters to an undisturbed run; (b) wrong sizes in both attempts -> every rank raises and parameters and Adam moments
    are bit-identical to their values before the step."""
    import numpy as np
    worker = os.path.join(REPO, "tests", "train_dist_worker.py")
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", LRT_CULL_BUILD="1")
    runs = {}
    for tag, extra, port in (("clean", {}, "29631"), ("once", {"LRT_TEST_BAD_CULL": "once:4"}, "29633"), ("always", {"LRT_TEST_BAD_CULL": "always:4"}, "29635")):
        subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", port,
                        worker, str(tmp_path / tag), "cuda:0", "sparse"], check=True, env=dict(base, **extra), cwd=REPO, timeout=600)
        runs[tag] = [np.load(str(tmp_path / f"{tag}.rank{r}.npz")) for r in range(2)]
    assert int(runs["clean"][0]["redone"][0]) == 0
    assert int(runs["once"][0]["redone"][0]) >= 1 and int(runs["once"][1]["redone"][0]) == int(runs["once"][0]["redone"][0])
    for k in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "m_xyz", "v_xyz"):
        np.testing.assert_array_equal(runs["once"][0][k], runs["once"][1][k])                     # replicas never diverge
    assert int(runs["always"][0]["raised_clean"][0]) == 1 and int(runs["always"][1]["raised_clean"][0]) == 1

    def same_run(a, b):       # two runs sum a Gaussian's per-hit gradients in the arrival order of LDS atomics: equal up to that rounding
        np.testing.assert_array_equal(a["log"][:, 1:], b["log"][:, 1:])                           # same P, same clone / split / prune counts
        for k in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"):
            x, y = a[k].astype(np.float64), b[k].astype(np.float64)
            # (a handful of elements may differ by a few learning rates: Adam turns a rounding-level sign change of a near-zero gradient into
            # a full step -- seen: 6 of 28,357 opacities apart by ~0.03, relative L2 2e-4; a step taken on an incomplete gradient moves thousands)
            far = np.abs(x - y) > 1e-5 * max(np.abs(y).max(), 1.0)
            assert far.mean() < 2e-3 and np.linalg.norm((x - y)[~far]) / np.linalg.norm(y) < 1e-4, k
    same_run(runs["once"][0], runs["clean"][0])                                                   # the redone step left no trace
    same_run(runs["always"][0], runs["clean"][0])                                                 # ... nor did the refused one


def test_culled_builds_survive_ray_sets_whose_kept_count_changes_severalfold():
    """A rank's sector keeps very different numbers of Gaussians from pose to pose (training frames drawn at random from a drive).  The
    culled build of a ray set seen before is sized from ITS OWN last count, a new one reads its count back (ShardedTracer._cull_sizing):
    alternating a narrow and a wide sector -- kept counts an order of magnitude apart -- must neither lose primitives (error code 8) nor
    change the results, and must read back exactly once per ray set."""
    sys.path.insert(0, REPO)
    import numpy as np
    from lidar_rt_amd import scenes
    from lidar_rt_amd.parallel import ShardedTracer
    dev = torch.device("cuda", 0)
    sc = scenes.make_scene(120_000, seed=5, radius_scale=0.4)
    o, d = scenes.kitti_rays(16, 512)
    t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
    bg = torch.as_tensor(scenes.BG_DEFAULT, device=dev)
    dL = torch.as_tensor(scenes.upstream_grad(16, 512), device=dev)
    args = (t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg)
    slabs = {"narrow": (0, 32), "wide": (128, 384)}
    rays = {k: (torch.as_tensor(o[:, a:b].copy(), device=dev), torch.as_tensor(d[:, a:b].copy(), device=dev), dL[:, a:b].contiguous()) for k, (a, b) in slabs.items()}
    ref = {}
    plain = ShardedTracer()
    for k, (ro, rd, g) in rays.items():
        out, _ = plain.forward(ro, rd, *args)
        ref[k] = (out.clone(), {n: v.clone() for n, v in plain.backward(*args, g).items()})
    tr = ShardedTracer(); tr.cull_build = True
    kept = {}
    for rep in range(3):
        for k, (ro, rd, g) in rays.items():
            out, _ = tr.forward(ro, rd, *args, cull_key=k)
            gr = tr.backward(*args, g)
            torch.cuda.synchronize()
            kept[k] = tr.backend.state.built_count(dev)
            assert torch.equal(out, ref[k][0]), (rep, k)
            assert torch.equal(gr["accum"] > 0, ref[k][1]["accum"] > 0), (rep, k)      # the exact touched set: no float order can hide a lost primitive
            for n in ("means", "shs", "opacities"):
                # a Gaussian's per-hit terms are summed in the arrival order of the bucket sort's LDS atomics: signed float32 terms in another order
                # (a few ulp of the LARGEST term, not of the sum).  A lost primitive changes its Gaussians' gradients by O(1) of their size.
                a_, b_ = gr[n], ref[k][1][n]
                assert torch.allclose(a_, b_, rtol=1e-4, atol=2e-6 * float(b_.abs().max())), (rep, k, n, float((a_ - b_).abs().max()), float(b_.abs().max()))
    tr.forward(rays["narrow"][0], rays["narrow"][1], *args, cull_key="narrow")     # the check of the last wide step happens here
    assert kept["wide"] > 3 * kept["narrow"], kept
    assert tr.cull_readbacks == 2, tr.cull_readbacks
