"""`python -m lidar_rt_amd.train --data DIR` (the train.py-shaped loop on a file-backed sequence) on the KITTI-360-dynamic shape: 66 x 1030
range images, a background and 8 rigid actors with a pose per frame (BASELINE configs[3]; tools/make_sequence.py renders the sequence).
50 iterations from disk, `python -m lidar_rt_amd.evaluate` on its checkpoints, then a second run resumed from the iteration-25 checkpoint."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path):
    return torch.load(path, map_location="cpu", weights_only=False)


def test_fifty_iterations_from_disk_and_a_resume_from_the_checkpoint(tmp_path):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_sequence
    from lidar_rt_amd import sequence, train, training
    data = str(tmp_path / "seq")
    meta = make_sequence.make("kitti360_dynamic", data, n_frames=6, scale=0.1)
    assert (meta["height"], meta["width"]) == (66, 1030) and meta["dynamic"] and meta["n_actors"] == 8
    seq = sequence.load_sequence(data, "cuda:0")
    valid = float(torch.stack([seq.frames.get_mask(f).float().mean() for f in seq.train_frames]).mean())
    assert 0.05 < valid <= 1.0, valid                                      # the rendered ground truth has returns (a tenth of the dataset's Gaussians: a sparse scene)
    common = ["--data", data, "--log-every", "5", "--save-every", "25", "--max-points", "60000", "--opt", "lambda_cd=0.01"]
    run = lambda out, extra: subprocess.run([sys.executable, "-m", "lidar_rt_amd.train", "--out", out] + common + extra, cwd=REPO, capture_output=True,
                                            text=True, timeout=1500)
    a = run(str(tmp_path / "a"), ["--iters", "50"])
    assert a.returncode == 0, a.stdout[-2000:] + a.stderr[-3000:]
    rows = [json.loads(l) for l in a.stdout.splitlines() if l.startswith("{")]
    assert rows[-1]["iteration"] == 50 and rows[-1]["loss"] < 0.8 * rows[0]["loss"], rows                    # it trains
    for it in (25, 50):
        assert os.path.exists(tmp_path / "a" / f"chkpnt{it}.pth")
    # the evaluation entry point on the same directory and the two checkpoints (eval.py:370-470): one JSON object with the reference's metric set
    # (renders masked with the ground truth's ray-hit mask, eval.py:184: after 25 iterations the predicted ray-drop still says "no return" everywhere,
    # and a frame without predicted points has no Chamfer distance)
    evs = {}
    for it in (25, 50):
        e = subprocess.run([sys.executable, "-m", "lidar_rt_amd.evaluate", "--data", data, "--ckpt", str(tmp_path / "a" / f"chkpnt{it}.pth"), "--frames", "all", "--use-gt-mask",
                            "--max-points", "60000", "--out", str(tmp_path / f"eval{it}.json")], cwd=REPO, capture_output=True, text=True, timeout=900)
        assert e.returncode == 0, e.stdout[-2000:] + e.stderr[-3000:]
        evs[it] = json.loads(open(tmp_path / f"eval{it}.json").read())
        assert evs[it]["iteration"] == it and len(evs[it]["frames"]) == 6 and set(evs[it]["mean"]) == {"depth", "intensity", "raydrop", "points"}
        for g, ms in evs[it]["mean"].items():
            assert all(np.isfinite(v) for v in ms.values()), (it, g, ms)
    assert evs[50]["mean"]["depth"]["rmse"] < evs[25]["mean"]["depth"]["rmse"], (evs[25]["mean"]["depth"], evs[50]["mean"]["depth"])      # 25 more iterations: closer
    params25, it25 = _load(tmp_path / "a" / "chkpnt25.pth")
    assert it25 == 25 and len(params25) == 9 and len(params25[0]) == 12                                  # background + 8 actors, the reference's 12-tuple
    # restoring the checkpoint reproduces every tensor and the optimizer state bit for bit
    opt = training.default_options()
    scene = sequence.scene_from_sequence(seq, max_points=60000, seed=0)
    scene.training_setup(opt)
    scene.restore(_load(tmp_path / "a" / "chkpnt25.pth")[0], opt)
    for g, saved in zip(scene.gaussians_assets, params25):
        back = g.capture()
        for i in (1, 2, 3, 4, 5, 6, 8, 9):
            assert torch.equal(back[i].detach().cpu(), saved[i].detach().cpu()), i
        for k, st in saved[10]["state"].items():
            for n in ("exp_avg", "exp_avg_sq"):
                assert torch.equal(back[10]["state"][k][n].cpu(), st[n].cpu())
    # a second process resumes at 26 and runs to 50: same frames, same densification draws; the step's float sums carry atomic-order noise
    b = run(str(tmp_path / "b"), ["--iters", "50", "--resume", str(tmp_path / "a" / "chkpnt25.pth")])
    assert b.returncode == 0, b.stdout[-2000:] + b.stderr[-3000:]
    rows_b = [json.loads(l) for l in b.stdout.splitlines() if l.startswith("{")]
    assert [r["iteration"] for r in rows_b] == [r["iteration"] for r in rows if r["iteration"] > 25]
    assert [r["frame"] for r in rows_b] == [r["frame"] for r in rows if r["iteration"] > 25]
    assert [r["points"] for r in rows_b] == [r["points"] for r in rows if r["iteration"] > 25]
    for ra, rb in zip([r for r in rows if r["iteration"] > 25], rows_b):
        assert abs(ra["loss"] - rb["loss"]) <= 5e-3 * abs(ra["loss"]), (ra, rb)
    pa, pb = _load(tmp_path / "a" / "chkpnt50.pth")[0], _load(tmp_path / "b" / "chkpnt50.pth")[0]
    for ga, gb in zip(pa, pb):
        for i, one_step in ((1, 1e-2), (4, 5e-3), (6, 5e-2)):              # positions, log-scales, opacity logits and the size of ONE Adam step of each at this stage
            # Adam (eps 1e-15, as in the reference) turns a gradient that is float-sum-order noise around zero into full steps of +-lr: such elements
            # walk apart (after 25 iterations 3 % of the positions and 13-30 % of the opacity logits differ by more than 2 % of a step); measured in
            # units of the step the two runs stay together: hardly any element is a whole step apart, the typical one a small fraction of a step --
            # and the loss of every iteration agrees (above)
            x, y = ga[i].detach().double(), gb[i].detach().double()
            apart = ((x - y).abs() > one_step).double().mean()
            assert float(apart) < 0.05, (i, float(apart))
            assert float((x - y).abs().median()) < 0.1 * one_step, (i, float((x - y).abs().median()))


def test_a_deterministic_run_resumed_from_its_checkpoint_repeats_the_uninterrupted_run_bit_for_bit(tmp_path):
    """--deterministic (Tracer(deterministic=True)): the gradient sums are taken in a fixed order and the forward keeps no learnt state, so a second
    process resumed from the iteration-12 checkpoint must arrive at the SAME iteration-24 checkpoint -- every parameter and Adam moment bit for bit,
    every logged loss equal, the same clone / split / prune counts in the densifications on both sides of the checkpoint -- as the run that was never
    interrupted (train.py:67-447's resume).  All loss terms at the reference's weights (the Chamfer term carries no gradient, as in
    the reference: lidar_sensor.py:182-183)."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_sequence
    data = str(tmp_path / "seq")
    make_sequence.make("kitti360_dynamic", data, n_frames=4, scale=0.1)
    common = ["--data", data, "--log-every", "1", "--save-every", "12", "--max-points", "60000", "--deterministic",
              "--opt", "densify_from_iter=4", "--opt", "densification_interval=8"]                  # densifications at 8, 16 and 24
    run = lambda out, extra: subprocess.run([sys.executable, "-m", "lidar_rt_amd.train", "--out", out] + common + extra, cwd=REPO, capture_output=True,
                                            text=True, timeout=1500)
    a = run(str(tmp_path / "a"), ["--iters", "24"])
    assert a.returncode == 0, a.stdout[-2000:] + a.stderr[-3000:]
    b = run(str(tmp_path / "b"), ["--iters", "24", "--resume", str(tmp_path / "a" / "chkpnt12.pth")])
    assert b.returncode == 0, b.stdout[-2000:] + b.stderr[-3000:]
    rows_a = [json.loads(l) for l in a.stdout.splitlines() if l.startswith("{")]
    rows_b = [json.loads(l) for l in b.stdout.splitlines() if l.startswith("{")]
    tail_a = [r for r in rows_a if r["iteration"] > 12]
    assert [r["iteration"] for r in rows_b] == [r["iteration"] for r in tail_a] and len(rows_b) == 12
    for ra, rb in zip(tail_a, rows_b):
        assert ra["frame"] == rb["frame"] and ra["points"] == rb["points"] and ra["loss"] == rb["loss"], (ra, rb)
    pa, pb = _load(tmp_path / "a" / "chkpnt24.pth")[0], _load(tmp_path / "b" / "chkpnt24.pth")[0]
    assert len(pa) == len(pb) == 9
    assert len({r["points"] for r in rows_a}) > 1, [r["points"] for r in rows_a]                    # the densifications changed the number of Gaussians
    for ga, gb in zip(pa, pb):
        for i in (1, 2, 3, 4, 5, 6, 8, 9):
            assert torch.equal(ga[i].detach().cpu(), gb[i].detach().cpu()), i
        for k, st in ga[10]["state"].items():
            for n in ("exp_avg", "exp_avg_sq"):
                assert torch.equal(gb[10]["state"][k][n].cpu(), st[n].cpu()), (k, n)
