"""The file-backed sequence layout (lidar_rt_amd/sequence.py) and the loop entry point's host logic (lidar_rt_amd/train.py): CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from lidar_rt_amd import sequence, scenes
from lidar_rt_amd.training import RangeFrames


def _frames(n, H=6, W=16, waymo=False, seed=0):
    r = np.random.default_rng(seed)
    out = []
    for f in range(n):
        depth = r.uniform(2, 50, (H, W)).astype(np.float32)
        mask = r.random((H, W)) > 0.2
        out.append({"id": 10 + 3 * f, "depth": depth * mask, "intensity": r.random((H, W)).astype(np.float32) * mask, "mask": mask,
                    "inclination": scenes.waymo_inclinations(H) if waymo else np.array([-0.43, 0.035], np.float32),
                    "sensor2world": scenes.pose_matrix((0.5 * f, 0.1, 0.2), yaw=0.02 * f, pitch=0.01)})
    return out


@pytest.mark.parametrize("waymo", [False, True])
def test_round_trip_gives_the_frames_the_rays_and_the_boxes(tmp_path, waymo):
    fr = _frames(4, waymo=waymo)
    A = 2
    boxes = {"frames": [f["id"] for f in fr], "translation": np.arange(A * 4 * 3, dtype=np.float32).reshape(A, 4, 3),
             "quaternion": np.tile(np.array([0.8, 0, 0, 0.6], np.float32), (A, 4, 1)), "size": np.array([[4, 2, 1.5], [5, 2.2, 2]], np.float32),
             "valid": np.array([[1, 1, 1, 1], [1, 0, 1, 1]], bool)}
    init = {"background": {"points": np.random.default_rng(1).normal(size=(50, 3)).astype(np.float32), "intensity": np.full(50, 0.3, np.float32)}}
    s2e = scenes.pose_matrix((1.4, 0, 2.1), yaw=0.02) if waymo else None
    meta = sequence.write_sequence(str(tmp_path), fr, data_type="Waymo" if waymo else "KITTI", extent=42.0, sensor2ego=s2e, boxes=boxes, init=init,
                                   test_frames=[fr[-1]["id"]])
    assert meta["frames"] == [10, 13, 16, 19] and meta["dynamic"] and meta["n_actors"] == 2
    assert json.load(open(tmp_path / "meta.json"))["format"] == sequence.FORMAT
    seq = sequence.load_sequence(str(tmp_path), device="cpu")
    assert seq.train_frames == [10, 13, 16] and seq.test_frames == [19]
    for f in fr:
        np.testing.assert_array_equal(seq.frames.get_depth(f["id"]).numpy(), f["depth"])
        np.testing.assert_array_equal(seq.frames.get_mask(f["id"]).numpy(), f["mask"])
        inc = [float(x) for x in f["inclination"]] if waymo else (float(f["inclination"][0]), float(f["inclination"][1]))
        o, d = RangeFrames.range_rays(6, 16, inc, torch.as_tensor(f["sensor2world"]), "Waymo" if waymo else "KITTI", None if s2e is None else torch.as_tensor(s2e))
        np.testing.assert_array_equal(seq.frames.get_range_rays(f["id"])[1].numpy(), d.numpy())       # the sensor model of the training loop
        o2, d2 = scenes.range_rays(6, 16, f["inclination"], f["sensor2world"], "Waymo" if waymo else "KITTI", s2e)
        assert np.abs(d.numpy() - d2).max() < 1e-6                                                     # ... which the golden ray grids pin
    assert len(seq.boxes) == 2 and sorted(seq.boxes[1].frame) == [10, 16, 19]                         # the invalid pose is absent
    t, q, _, _ = seq.boxes[0].frame[13]
    np.testing.assert_array_equal(t.numpy(), boxes["translation"][0, 1]); assert tuple(q.shape) == (1, 4)
    np.testing.assert_allclose(seq.boxes[1].max_xyz.numpy(), [2.5, 1.1, 1.0])
    np.testing.assert_array_equal(seq.init["background"]["points"].numpy(), init["background"]["points"])


def test_loader_refuses_what_it_cannot_trust(tmp_path):
    fr = _frames(2)
    sequence.write_sequence(str(tmp_path), fr)
    os.remove(tmp_path / "frames" / "000013.npz")
    with pytest.raises(FileNotFoundError, match="frame 13"):
        sequence.load_sequence(str(tmp_path), device="cpu")
    meta = json.load(open(tmp_path / "meta.json")); meta["format"] = "something-else/9"
    json.dump(meta, open(tmp_path / "meta.json", "w"))
    with pytest.raises(ValueError, match="not a lidar-rt-amd-sequence/1"):
        sequence.load_sequence(str(tmp_path), device="cpu")
    bad = _frames(2); bad[1]["depth"] = bad[1]["depth"][:, :8]
    with pytest.raises(ValueError, match="differs from the sequence's"):
        sequence.write_sequence(str(tmp_path / "b"), bad)
    bad = _frames(1); bad[0]["inclination"] = np.zeros(5, np.float32)
    with pytest.raises(ValueError, match="inclination"):
        sequence.write_sequence(str(tmp_path / "c"), bad)
    with pytest.raises(ValueError, match="at least one frame"):
        sequence.write_sequence(str(tmp_path / "d"), [])


def test_frame_order_is_a_function_of_seed_and_iteration_and_the_entry_point_refuses_a_short_node(tmp_path, capsys):
    from lidar_rt_amd import train
    frames = [3, 5, 8, 13]
    a = [train.frame_of(7, it, frames) for it in range(1, 200)]
    assert a == [train.frame_of(7, it, frames) for it in range(1, 200)] and set(a) == set(frames)
    assert a != [train.frame_of(8, it, frames) for it in range(1, 200)]
    if torch.cuda.is_available():
        pytest.skip("GPU present: the refusals below are the no-device paths")
    assert train.main(["--data", str(tmp_path), "--iters", "3", "--gpus", "4"]) == 2                  # fewer devices than ranks
    assert "refusing" in capsys.readouterr().err
    with pytest.raises(SystemExit, match="needs a HIP device"):
        train.main(["--data", str(tmp_path), "--iters", "3"])


def test_the_entry_point_refuses_deterministic_with_exact_accum_and_the_tracer_flag_implies_deferred_weights(tmp_path, capsys):
    """--deterministic takes the hit weights from the backward (the forward's are float atomics in arrival order): together with --exact-accum it is refused
    before any device is touched; `Tracer(deterministic=True)` implies `deferred_accum`."""
    from lidar_rt_amd import train
    with pytest.raises(SystemExit) as e:
        train.main(["--data", str(tmp_path), "--iters", "3", "--deterministic", "--exact-accum"])
    assert e.value.code == 2 and "--deterministic" in capsys.readouterr().err
    from lidar_rt_amd.diff_lidar_tracer import Tracer
    t = Tracer(deterministic=True)
    assert t.deterministic and t.deferred_accum
    assert not Tracer().deterministic and not Tracer().deferred_accum
