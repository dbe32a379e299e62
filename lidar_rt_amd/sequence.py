"""File-backed LiDAR sequences for the training / evaluation loops ("a range-image dataset adapter").

The reference reads Waymo tfrecords / KITTI-360 raw files into a ``LiDARSensor`` (lib/dataloader/waymo_loader, kitti_loader; range
images, per-beam inclinations, sensor poses) and its actors' tracking boxes into ``BoundingBox`` objects (lib/dataloader/gs_loader.py:
220-298).  Those readers (tensorflow, open3d) are out of scope; what the loops need is the data they produce.  This module defines a
NEUTRAL on-disk layout for exactly that data -- so that a converter from any dataset is a few lines of numpy -- and reads it into the
in-memory ``RangeFrames`` / ``GaussianScene`` of ``lidar_rt_amd.training``:

    DIR/meta.json                 {"format": "lidar-rt-amd-sequence/1", "data_type": "KITTI" | "Waymo", "height": H, "width": W,
                                   "frames": [ids...], "test_frames": [ids...], "dynamic": bool, "extent": metres, "sensor2ego": 4x4 | null}
    DIR/frames/000123.npz         depth (H,W) f32 [m, 0 = no return] | intensity (H,W) f32 | mask (H,W) bool [valid return]
                                  | inclination (2,) bounds or (H,) per-beam table [rad] | sensor2world (4,4) f32
    DIR/boxes.npz      (optional) frames (F,) i64 | translation (A,F,3) f32 | quaternion (A,F,4) f32 (w,x,y,z: actor -> world)
                                  | valid (A,F) bool | size (A,3) f32 [tracking-box extents, metres]
    DIR/init/background.npz, DIR/init/actor_00.npz ...   (optional) points (N,3) f32 [world / actor frame] | intensity (N,) f32
                                  | normals (N,3) f32 (optional): the initial point clouds (gaussian_model.py:155-184); without them the
                                  background is initialised from the back-projected returns of the training frames, an actor from
                                  random points in its box.

``write_sequence`` is the writer (tools/make_sequence.py renders synthetic sequences of the BASELINE configs' shapes with it);
``load_sequence`` the reader; ``scene_from_sequence`` builds the Gaussian assets.  ``python -m lidar_rt_amd.train --data DIR`` is the loop.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

from .training import GaussianAsset, GaussianScene, RangeFrames

FORMAT = "lidar-rt-amd-sequence/1"


class TrackingBox:
    """What the loops read of the reference's BoundingBox (lib/scene/bounding_box.py): per-frame pose ``frame[ts] = (translation (3,),
    quaternion (1,4), None, None)`` (actor -> world) and the box's extent in the actor frame (``min_xyz`` / ``max_xyz``)."""

    def __init__(self, size, device):
        half = 0.5 * torch.as_tensor(np.asarray(size, np.float32), device=device)
        self.min_xyz, self.max_xyz = -half, half
        self.frame: Dict[int, tuple] = {}


def _frame_path(root: str, fid: int) -> str:
    return os.path.join(root, "frames", f"{int(fid):06d}.npz")


def write_sequence(root: str, frames: Iterable[dict], data_type: str = "KITTI", extent: float = 1.0, sensor2ego=None, boxes: Optional[dict] = None,
                   init: Optional[Dict[str, dict]] = None, test_frames: Sequence[int] = ()) -> dict:
    """frames: dicts with id, depth, intensity, mask, inclination, sensor2world (numpy / torch).  boxes / init: see the module docstring.
    Returns the meta dictionary it wrote."""
    os.makedirs(os.path.join(root, "frames"), exist_ok=True)
    ids, H, W = [], None, None
    npy = lambda a, dt=np.float32: np.ascontiguousarray((a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)).astype(dt))
    for fr in frames:
        d = npy(fr["depth"])
        if H is None:
            H, W = d.shape
        if d.shape != (H, W):
            raise ValueError(f"frame {fr['id']}: range image {d.shape} differs from the sequence's {(H, W)}")
        inc = npy(fr["inclination"]).reshape(-1)
        if inc.size not in (2, H):
            raise ValueError(f"frame {fr['id']}: inclination must hold 2 bounds or {H} per-beam angles, got {inc.size}")
        np.savez(_frame_path(root, fr["id"]), depth=d, intensity=npy(fr["intensity"]).reshape(H, W), mask=npy(fr["mask"], np.bool_).reshape(H, W),
                 inclination=inc, sensor2world=npy(fr["sensor2world"]).reshape(4, 4))
        ids.append(int(fr["id"]))
    if not ids:
        raise ValueError("a sequence needs at least one frame")
    if boxes is not None:
        A = np.asarray(boxes["translation"]).shape[0]
        F_ = len(boxes["frames"])
        np.savez(os.path.join(root, "boxes.npz"), frames=np.asarray(boxes["frames"], np.int64), translation=npy(boxes["translation"]).reshape(A, F_, 3),
                 quaternion=npy(boxes["quaternion"]).reshape(A, F_, 4), valid=npy(boxes.get("valid", np.ones((A, F_), bool)), np.bool_).reshape(A, F_),
                 size=npy(boxes["size"]).reshape(A, 3))
    if init:
        os.makedirs(os.path.join(root, "init"), exist_ok=True)
        for name, cloud in init.items():
            arrs = {"points": npy(cloud["points"]).reshape(-1, 3), "intensity": npy(cloud["intensity"]).reshape(-1)}
            if cloud.get("normals") is not None:
                arrs["normals"] = npy(cloud["normals"]).reshape(-1, 3)
            np.savez(os.path.join(root, "init", name + ".npz"), **arrs)
    meta = {"format": FORMAT, "data_type": data_type, "height": int(H), "width": int(W), "frames": sorted(ids),
            "test_frames": sorted(int(t) for t in test_frames), "dynamic": boxes is not None, "extent": float(extent),
            "n_actors": int(np.asarray(boxes["translation"]).shape[0]) if boxes is not None else 0,
            "sensor2ego": None if sensor2ego is None else np.asarray(sensor2ego, np.float64).reshape(4, 4).tolist()}
    with open(os.path.join(root, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    return meta


def load_sequence(root: str, device="cuda", frames: Optional[Sequence[int]] = None) -> SimpleNamespace:
    """-> namespace(meta, frames: RangeFrames, boxes: [TrackingBox], init: {name: {points, intensity, normals}}, train_frames, test_frames).
    The ray grids are derived from inclination + pose like ``LiDARSensor.get_range_rays`` does (RangeFrames.range_rays)."""
    with open(os.path.join(root, "meta.json")) as f:
        meta = json.load(f)
    if meta.get("format") != FORMAT:
        raise ValueError(f"{root}: not a {FORMAT} directory (format = {meta.get('format')!r})")
    dev = torch.device(device)
    rf = RangeFrames()
    s2e = None if meta.get("sensor2ego") is None else torch.tensor(meta["sensor2ego"], dtype=torch.float32, device=dev)
    want = list(meta["frames"]) if frames is None else [int(f) for f in frames]
    for fid in want:
        p = _frame_path(root, fid)
        if not os.path.exists(p):
            raise FileNotFoundError(f"{root}: meta.json lists frame {fid} but {p} is missing")
        z = np.load(p)
        d = torch.as_tensor(z["depth"], device=dev)
        if tuple(d.shape) != (meta["height"], meta["width"]):
            raise ValueError(f"{p}: range image {tuple(d.shape)} differs from meta.json's {(meta['height'], meta['width'])}")
        inc = [float(x) for x in z["inclination"].reshape(-1)]
        rf.add_range_image(fid, d, torch.as_tensor(z["intensity"], device=dev), torch.as_tensor(z["mask"], device=dev),
                           inc if len(inc) > 2 else (inc[0], inc[1]), torch.as_tensor(z["sensor2world"], device=dev), meta["data_type"], s2e)
    boxes: List[TrackingBox] = []
    bp = os.path.join(root, "boxes.npz")
    if os.path.exists(bp):
        z = np.load(bp)
        fids = [int(x) for x in z["frames"]]
        for a in range(z["translation"].shape[0]):
            tb = TrackingBox(z["size"][a], dev)
            for k, fid in enumerate(fids):
                if z["valid"][a, k]:
                    tb.frame[fid] = (torch.as_tensor(z["translation"][a, k], device=dev), torch.as_tensor(z["quaternion"][a, k], device=dev).reshape(1, 4), None, None)
            boxes.append(tb)
    init = {}
    ip = os.path.join(root, "init")
    if os.path.isdir(ip):
        for fn in sorted(os.listdir(ip)):
            if fn.endswith(".npz"):
                z = np.load(os.path.join(ip, fn))
                init[fn[:-4]] = {k: torch.as_tensor(z[k], device=dev) for k in z.files}
    test = [t for t in meta.get("test_frames", []) if t in rf.rays]
    train = [f for f in sorted(rf.rays) if f not in set(test)] or sorted(rf.rays)
    return SimpleNamespace(meta=meta, frames=rf, boxes=boxes, init=init, train_frames=train, test_frames=test, root=root)


def scene_from_sequence(seq: SimpleNamespace, max_sh_degree: int = 3, max_points: int = 2_000_000, seed: int = 0) -> GaussianScene:
    """Gaussian assets for a loaded sequence (asset 0 = background, then one per tracking box; gs_loader.py:88-160): from DIR/init/*.npz
    where present, else the background from the back-projected valid returns of the training frames (sub-sampled to ``max_points``,
    intensity as the DC colour) and an actor from 2000 random points in its box."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    rf, dev = seq.frames, next(iter(seq.frames.depth.values())).device
    extent = float(seq.meta.get("extent", 1.0))
    if "background" in seq.init:
        c = seq.init["background"]
        pts, inten, nrm = c["points"], c["intensity"], c.get("normals")
    else:
        ps, it = [], []
        for f in seq.train_frames:
            ps.append(rf.inverse_projection_with_range(f, rf.get_depth(f)))
            it.append(rf.get_intensity(f).reshape(-1).index_select(0, rf.mask_index[f]))
        pts, inten, nrm = torch.cat(ps), torch.cat(it), None
        if pts.shape[0] > max_points:
            sel = torch.randperm(pts.shape[0], generator=g)[:max_points].to(dev)
            pts, inten = pts[sel], inten[sel]
    assets = [GaussianAsset.from_points(pts, inten.clamp(0, 1), nrm, max_sh_degree=max_sh_degree, extent=extent)]
    for a, tb in enumerate(seq.boxes):
        name = f"actor_{a:02d}"
        if name in seq.init:
            c = seq.init[name]
            p_, i_, n_ = c["points"], c["intensity"], c.get("normals")
        else:
            u = torch.rand((2000, 3), generator=g).to(dev)
            p_, i_, n_ = tb.min_xyz + u * (tb.max_xyz - tb.min_xyz), torch.full((2000,), 0.5, device=dev), None
        assets.append(GaussianAsset.from_points(p_, i_.clamp(0, 1), n_, max_sh_degree=max_sh_degree, bounding_box=tb,
                                                extent=float((tb.max_xyz - tb.min_xyz).norm())))
    return GaussianScene(assets)
