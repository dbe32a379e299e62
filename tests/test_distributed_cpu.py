"""N > 1 path on CPU: azimuth sharding + slab all_gather + fused gradient all_reduce over gloo (world_size 2 and 3),
with an oracle-backed local tracer, must reproduce the single-process result."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from lidar_rt_amd import scenes
from lidar_rt_amd.parallel import ShardedTracer, column_slab, GradLayout
from tests.oracle_backend import OracleBackend

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_column_slabs_partition_the_image():
    for W in (1, 7, 45, 2048, 2650):
        for n in (1, 2, 3, 8):
            edges = [column_slab(W, r, n) for r in range(n)]
            assert edges[0][0] == 0 and edges[-1][1] == W
            assert all(edges[i][1] == edges[i + 1][0] for i in range(n - 1))
            widths = [b - a for a, b in edges]
            assert max(widths) - min(widths) <= 1


def test_grad_layout_is_one_flat_buffer():
    lay = GradLayout(10, 16, "cpu")
    assert lay.flat.numel() == 10 * (3 + 2 + 4 + 1 + 48) + 10
    lay.flat.zero_()
    lay.views["shs"].fill_(2.0); lay.views["accum"].fill_(3.0)
    assert float(lay.flat.sum()) == 10 * 48 * 2.0 + 10 * 3.0


@pytest.mark.parametrize("world,exchange", [(2, "dense"), (2, "sparse"), (3, "auto"), (2, "auto-dense"), (2, "owner"), (3, "owner")])
def test_sharded_equals_single(tmp_path, world, exchange):
    sc = scenes.make_scene(1500, seed=4, radius_scale=0.2)
    o, d = scenes.kitti_rays(6, 45)
    dL = scenes.upstream_grad(6, 45)
    t = {k: torch.from_numpy(v) for k, v in sc.items()}
    bg = torch.tensor([0.0, 0.0, 1.0])
    single = ShardedTracer(backend=OracleBackend())
    out1, acc1 = single.forward(torch.from_numpy(o), torch.from_numpy(d), t["means"], t["scales"], t["rotations"],
                                t["opacities"], t["shs"], 3, bg)
    g1 = single.backward(t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg, torch.from_numpy(dL))
    base = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + world), os.path.join(REPO, "tests", "dist_worker.py"), base, exchange]
    subprocess.run(cmd, check=True, env=env, cwd=REPO, timeout=600)
    res0 = np.load(base + ".rank0.npz")
    if exchange == "owner":
        # every Gaussian's gradient is complete on its owner (and only there); all ranks agree on the owner map
        owners = res0["owner"]
        assert set(np.unique(owners)) <= set(range(world)) and len(np.unique(owners)) == world
        for r in range(world):
            res = np.load(base + f".rank{r}.npz")
            np.testing.assert_array_equal(res["owner"], owners)
            np.testing.assert_allclose(res["out"], out1.numpy(), rtol=1e-6, atol=1e-7)
            mine = owners == r
            for k in ("means", "scales", "rotations", "opacities", "shs", "accum"):
                ref = g1[k].numpy()
                np.testing.assert_allclose(res[k][mine], ref[mine], rtol=1e-4, atol=1e-6 * max(np.abs(ref).max(), 1e-30))
        return
    for r in range(world):
        res = np.load(base + f".rank{r}.npz")
        for k in ("means", "scales", "rotations", "opacities", "shs", "accum"):
            np.testing.assert_array_equal(res[k], res0[k])                                    # replicas stay bit-identical
        np.testing.assert_allclose(res["out"], out1.numpy(), rtol=1e-6, atol=1e-7)          # every rank sees the whole image
        for k in ("means", "scales", "rotations", "opacities", "shs", "accum"):
            ref = g1[k].numpy()
            np.testing.assert_allclose(res[k], ref, rtol=1e-4, atol=1e-6 * max(np.abs(ref).max(), 1e-30))


def _run_training_worker(tmp_path, tag, world, device="cpu", exchange="sparse", port=29560, **env_extra):
    base = str(tmp_path / tag)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", **env_extra)
    worker = os.path.join(REPO, "tests", "train_dist_worker.py")
    if world == 1:
        env.pop("RANK", None); env.pop("WORLD_SIZE", None)
        cmd = [sys.executable, worker, base, device, exchange]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port + world), worker, base, device, exchange]
    subprocess.run(cmd, check=True, env=env, cwd=REPO, timeout=900)
    return [np.load(base + f".rank{r}.npz") for r in range(world)]


PARAMS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")


@pytest.mark.parametrize("world,exchange,first_cap", [(2, "sparse", None), (3, "sparse", None), (2, "dense", None), (3, "sparse", "40")])
def test_sharded_training_steps_equal_the_single_rank_run(tmp_path, world, exchange, first_cap):
    """K = 5 `training_step`s incl. one densification (clones, splits, opacity pruning) through `renderer.sharded`
    on N gloo ranks: every rank ends with the single-rank parameters (<= 1e-6), the same number of Gaussians, and the replicas are
    bit-identical to each other (losses, Adam and the densification decisions run replicated on replicated gradients).  With a first
    capacity that is too small the exchange verifies its counts inside the step and runs again: same result."""
    single = _run_training_worker(tmp_path, "single", 1)[0]
    assert single["log"][2, 2] > 0 and single["log"][2, 3] > 0 and single["log"][2, 5] > 0, "the case must clone, split and prune"
    extra = {"LRT_TEST_FIRST_CAP": first_cap} if first_cap else {}
    ranks = _run_training_worker(tmp_path, f"w{world}", world, exchange=exchange, **extra)
    for r, res in enumerate(ranks):
        np.testing.assert_array_equal(res["log"][:, 1:], single["log"][:, 1:])               # P and the clone / split / prune counts per step
        for k in PARAMS + ("m_xyz", "v_xyz"):
            assert res[k].shape == single[k].shape, (r, k)
            np.testing.assert_array_equal(res[k], ranks[0][k])                                 # replicas never diverge
            np.testing.assert_allclose(res[k], single[k], rtol=0, atol=1e-6 * max(np.abs(single[k]).max(), 1.0))
        if first_cap:
            assert int(res["reruns"][0]) >= 1, "the undersized first exchange must have been re-run"


def test_culled_builds_are_sized_per_ray_set():
    """ShardedTracer._cull_sizing (the policy, against a stand-in for the library state): a ray set seen for the first time reads its
    count back (cull_next = 0), a known one is sized from ITS OWN last count x 1.25 + 4096 whatever the build in between kept, and the
    table forgets the oldest sets first."""
    import torch
    from lidar_rt_amd.parallel import ShardedTracer

    class State:
        def __init__(self): self.last, self.next = -1, None
        def get_option(self, name, dev=None): assert name == "cull_last"; return self.last
        def set_option(self, name, v): assert name == "cull_next"; self.next = v

    class Backend:
        def __init__(self): self.state = State()

    tr = ShardedTracer(backend=Backend())
    tr._dev = torch.device("cpu")
    st = tr.backend.state
    tr._cull_sizing("A"); assert st.next == 0 and tr.cull_readbacks == 1          # unknown: read back
    st.last = 100_000                                                              # ... that build kept 100 k
    tr._cull_sizing("B"); assert st.next == 0 and tr.cull_readbacks == 2          # another set: read back, and A's count is learnt
    st.last = 900_000
    tr._cull_sizing("A"); assert st.next == 100_000 + 25_000 + 4096               # A again: its own count, not B's 900 k
    st.last = 101_000
    tr._cull_sizing("B"); assert st.next == 900_000 + 225_000 + 4096
    tr._cull_sizing("A"); assert st.next == 101_000 + 25_250 + 4096               # refreshed by its last build
    assert tr.cull_readbacks == 2
    for i in range(17000):                                                         # the table is bounded: the oldest sets go first
        tr._cull_counts[("x", i)] = 1
    st.last = 5
    tr._cull_sizing("C")
    assert len(tr._cull_counts) <= 16384 and "B" not in tr._cull_counts


def test_slab_balancer_moves_edges_towards_equal_time_and_keeps_the_tiling():
    """ShardedTracer._rebalance (pure arithmetic on the gathered times): widths stay multiples of the 8-column tile, every rank keeps at
    least one tile, the slow rank's slab shrinks and the fast one's grows, equal times leave equal widths alone, and the same inputs give the
    same edges (every rank runs this on the same gathered numbers)."""
    from lidar_rt_amd.parallel import ShardedTracer, column_slab

    def balancer(world, W):
        b = ShardedTracer.__new__(ShardedTracer); b.world = world; b.rank = 0
        b._edges = [column_slab(W, r, world)[0] for r in range(world)] + [W]; b._edges_key = (W, world); b._times = None
        return b

    W = 2048
    b = balancer(4, W); b._times = [1.0, 1.0, 1.0, 1.0]; b._rebalance(W)
    assert b._edges == [0, 512, 1024, 1536, 2048]
    b._times = [2.0, 1.0, 1.0, 1.0]; b._rebalance(W)
    w = [b._edges[i + 1] - b._edges[i] for i in range(4)]
    assert w[0] < 512 and all(x > 512 for x in w[1:]) and sum(w) == W and all(x % 8 == 0 for x in w)
    b2 = balancer(4, W); b2._times = [1.0, 1.0, 1.0, 1.0]; b2._rebalance(W); b2._times = [2.0, 1.0, 1.0, 1.0]; b2._rebalance(W)
    assert b2._edges == b._edges                                               # deterministic
    for _ in range(40):                                                         # a rank that stays 50x slower is squeezed, never below one tile
        b._times = [50.0, 1.0, 1.0, 1.0]; b._rebalance(W)
    w = [b._edges[i + 1] - b._edges[i] for i in range(4)]
    assert w[0] >= 8 and min(w) >= 8 and sum(w) == W and all(x % 8 == 0 for x in w)
    b = balancer(8, 2650)                                                       # a width that is not a multiple of 8 x ranks (configs[4])
    b._times = [1.0, 1.2, 1.3, 1.1, 0.9, 1.6, 1.4, 1.1]; b._rebalance(2650)
    e = b._edges
    assert e[0] == 0 and e[-1] == 2650 and all(e[i + 1] > e[i] for i in range(8)) and all(x % 8 == 0 for x in e[1:-1])
    b._times = [1.0, 0.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]; before = list(b._edges); b._rebalance(2650)
    assert b._edges == before                                                   # a missing time (0) changes nothing


def test_a_moved_slab_edge_does_not_cost_a_read_back():
    """After the balancer has moved a rank's edges by a few tiles the culled build of the new slab is sized from the old slab's count scaled
    by the width (x 1.15, then the usual x 1.25 + 4096) -- no host wait; a slab that overlaps less than 75 % is unknown."""
    import torch
    from lidar_rt_amd.parallel import ShardedTracer

    class State:
        def __init__(self): self.last, self.next = -1, None
        def get_option(self, name, dev=None): return self.last
        def set_option(self, name, v): self.next = v

    class Backend:
        def __init__(self): self.state = State()

    tr = ShardedTracer(backend=Backend()); tr._dev = torch.device("cpu"); st = tr.backend.state
    tr._cull_sizing(("rays", 256, 512, 1000)); assert st.next == 0
    st.last = 100_000
    tr._cull_sizing(("rays", 248, 520, 1000))                                   # 272 columns instead of 256, same rays and P
    g = int(100_000 * 272 / 256 * 1.15); assert st.next == g + g // 4 + 4096 and tr.cull_readbacks == 1
    st.last = 107_000
    tr._cull_sizing(("rays", 1024, 1280, 1000)); assert st.next == 0 and tr.cull_readbacks == 2      # another sector: unknown
    tr._cull_sizing(("rays", 256, 512, 1001)); assert st.next == 0                                   # another P: unknown
    tr._cull_sizing(("other", 256, 512, 1000)); assert st.next == 0                                  # other rays: unknown


# ---------------------------------------------------------------------------------- the step is verified before the optimizer moves
def _tiny_training_case():
    from lidar_rt_amd import training, renderer
    sc = scenes.make_scene(400, seed=4, radius_scale=0.2)
    t = lambda a: torch.as_tensor(a)
    asset = training.GaussianAsset.from_tensors(t(sc["means"]), t(sc["shs"][:, :1]).contiguous(), t(sc["shs"][:, 1:]).contiguous(),
                                                torch.log(t(sc["scales"])), t(sc["rotations"]), training.inverse_sigmoid(t(sc["opacities"])),
                                                max_sh_degree=3, extent=8.0)
    asset.active_sh_degree = 3
    scene = training.GaussianScene([asset])
    opt = training.default_options(); opt.lambda_cd = 0.0
    scene.training_setup(opt)
    frames = training.RangeFrames()
    rng = np.random.default_rng(7)
    H, W = 4, 24
    o, d = scenes.range_rays(H, W, (np.radians(-24.9), np.radians(2.0)), scenes.pose_matrix((0.0, 0.0, 0.0), yaw=0.0), "KITTI")
    frames.add_frame(0, t(o), t(d), t((4.0 + 0.3 * rng.standard_normal((H, W))).astype(np.float32)),
                     t(np.clip(0.5 + 0.2 * rng.standard_normal((H, W)), 0, 1).astype(np.float32)), t(rng.uniform(size=(H, W)) < 0.8))
    return training, renderer, scene, asset, opt, frames, t(scenes.BG_DEFAULT)


def test_training_step_redoes_a_step_whose_status_words_report_a_problem():
    """training_step asks ShardedTracer.verify_step() between loss.backward() and scene.optimize: one reported problem -> the gradients of
    that attempt are dropped and the step runs again; the parameters end where an undisturbed step puts them."""
    training, renderer, scene, asset, opt, frames, bg = _tiny_training_case()
    old_fused, old_sh = renderer.use_fused_preprocess, renderer.sharded
    try:
        renderer.use_fused_preprocess = False
        tr = ShardedTracer(backend=OracleBackend())
        renderer.sharded = tr
        r0 = training.training_step(scene, frames, 0, 1, opt, bg)
        assert r0["step_redone"] == 0
        ref = {n: p.detach().clone() for n, p in asset._params().items()}
        # the same step on a fresh copy of the scene, with a problem reported by the first verification
        training2, _, scene2, asset2, opt2, frames2, bg2 = _tiny_training_case()
        calls = []
        def fake_verify():
            calls.append(1)
            return ("forward", [0.0, 8.0]) if len(calls) == 1 else None
        tr.verify_step = fake_verify
        r1 = training2.training_step(scene2, frames2, 0, 1, opt2, bg2)
        assert r1["step_redone"] == 1 and len(calls) == 2
        for n, p in asset2._params().items():
            assert torch.equal(p.detach(), ref[n]), n              # the dropped attempt left nothing behind (gradients would have been added twice)
    finally:
        renderer.use_fused_preprocess, renderer.sharded = old_fused, old_sh


def test_training_step_refuses_the_optimizer_step_after_two_incomplete_attempts():
    from lidar_rt_amd._capi import LrtError
    training, renderer, scene, asset, opt, frames, bg = _tiny_training_case()
    old_fused, old_sh = renderer.use_fused_preprocess, renderer.sharded
    try:
        renderer.use_fused_preprocess = False
        tr = ShardedTracer(backend=OracleBackend())
        renderer.sharded = tr
        training.training_step(scene, frames, 0, 1, opt, bg)          # one good step: the Adam moments exist
        before = {n: p.detach().clone() for n, p in asset._params().items()}
        m_before = asset.optimizer.state[asset._xyz]["exp_avg"].clone()
        tr.verify_step = lambda: ("exchange", [1.0])
        with pytest.raises(LrtError, match="NOT taken"):
            training.training_step(scene, frames, 0, 2, opt, bg)
        for n, p in asset._params().items():
            assert torch.equal(p.detach(), before[n]), n
        assert torch.equal(asset.optimizer.state[asset._xyz]["exp_avg"], m_before)
    finally:
        renderer.use_fused_preprocess, renderer.sharded = old_fused, old_sh


def test_verify_step_consumes_pending_status_words_without_raising():
    """verify_step() = check() that returns instead of raising; an error bit 8 (a culled build lost primitives) makes the next culled build
    exact, an exchange overflow raises the capacity from the TRUE list lengths that travelled with the flag."""
    tr = ShardedTracer(backend=OracleBackend())
    assert tr.verify_step() is None
    tr._cull_prev_key = ("frame7", 0, 8, 100); tr._cull_counts[tr._cull_prev_key] = 123
    tr._pending.append(("forward", torch.tensor([0.0, 8.0, 0.0]), None))
    bad = tr.verify_step()
    assert bad == ("forward", [0.0, 8.0, 0.0]) and tr._cull_exact_next and ("frame7", 0, 8, 100) not in tr._cull_counts
    assert tr.verify_step() is None                                   # consumed
    tr._pending.append(("exchange", torch.tensor([1, 50000, 70000], dtype=torch.int32), None))
    assert tr.verify_step() == ("exchange", [1])
    assert tr._cap_hist[-1] == 70000 and tr._flat_dirty and tr.exchange_reruns == 1


# ---------------------------------------------------------------------------------- bench.py starts its own ranks
def test_bench_gpus_n_launches_n_ranks_or_refuses():
    import json
    bench = os.path.join(REPO, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, bench, "--gpus", "4", "--steps", "3", "--print-launch"], env=env, cwd=REPO, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and "--print-launch" not in cmd
    # no GPU here: four ranks cannot get a device each -> a refusal with a non-zero exit code, not a one-rank measurement
    out = subprocess.run([sys.executable, bench, "--gpus", "4"], env=env, cwd=REPO, capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "refusing" in out.stderr and not out.stdout.strip()
    # a torch.distributed environment of another size than --gpus is refused as well (the line would mislabel the run)
    out = subprocess.run([sys.executable, bench, "--gpus", "2"], env=dict(env, WORLD_SIZE="1", RANK="0"), cwd=REPO, capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "refusing" in out.stderr
