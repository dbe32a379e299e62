#!/usr/bin/env python3
"""CPU experiment (no GPU): how many BVH entries does a 16-ray tile visit under different primitive orders?

The collect kernel tests every quad of a visited leaf against all 16 rays of the tile, and 96.6 % of those tests miss
(profiles/r02_summary.md).  A leaf's box is the union of 8 quad boxes, so ONE large quad inflates it; this script builds the
implicit 8-wide tree of lrt_build.inc in numpy for several sort keys and counts, for a sample of tiles, the leaves / nodes whose
box any ray of the tile enters (whole ray, no slabs, no termination: an upper bound the kernel's counters scale with).
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidar_rt_amd import scenes

LEAF = int(os.environ.get("LEAF", 8))


def quat_R(q):
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)


def expand21(v):
    v = v.astype(np.uint64) & np.uint64(0x1fffff)
    v = (v | (v << np.uint64(32))) & np.uint64(0x1f00000000ffff)
    v = (v | (v << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
    v = (v | (v << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
    v = (v | (v << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
    v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
    return v


def morton(means):
    lo = means.min(0); ext = (means.max(0) - lo).max()
    c = np.clip((means - lo) * (2097151.0 / ext), 0, 2097151).astype(np.uint32)
    return (expand21(c[:, 0]) << np.uint64(2)) | (expand21(c[:, 1]) << np.uint64(1)) | expand21(c[:, 2])


def tree(lo, hi, order):
    """levels[0] = leaf boxes, levels[l] = boxes of level-l nodes' ... returns list of (lo, hi) per level (children grouping of 8)."""
    lo = lo[order]; hi = hi[order]
    n = len(lo); pad = (-n) % LEAF
    if pad:
        lo = np.concatenate([lo, np.full((pad, 3), 1e30)]); hi = np.concatenate([hi, np.full((pad, 3), -1e30)])
    L = [(lo.reshape(-1, LEAF, 3).min(1), hi.reshape(-1, LEAF, 3).max(1))]
    while len(L[-1][0]) > 1:
        a, b = L[-1]; n = len(a); pad = (-n) % 8
        if pad:
            a = np.concatenate([a, np.full((pad, 3), 1e30)]); b = np.concatenate([b, np.full((pad, 3), -1e30)])
        L.append((a.reshape(-1, 8, 3).min(1), b.reshape(-1, 8, 3).max(1)))
    return L


def tile_hits(blo, bhi, o, d):
    """bool (T, B): some of the tile's 16 rays enters box b.  o (3,), d (T, 16, 3)."""
    T = d.shape[0]
    out = np.zeros((T, len(blo)), bool)
    inv = 1.0 / np.where(np.abs(d) < 1e-30, 1e-30, d)
    for t in range(T):
        t0 = (blo[None] - o) * inv[t][:, None, :]; t1 = (bhi[None] - o) * inv[t][:, None, :]
        tn = np.minimum(t0, t1).max(2); tf = np.maximum(t0, t1).min(2)
        out[t] = ((tf >= tn) & (tf >= 0)).any(0)
    return out


def main():
    sc, ro, rd = scenes.s1m()
    mu = sc["means"].astype(np.float64); s = sc["scales"].astype(np.float64); op = sc["opacities"][:, 0].astype(np.float64)
    R = quat_R(sc["rotations"].astype(np.float64))
    f = np.sqrt(2 * np.log(op * 255.0)) + 0.01
    ex, ey = s[:, 0] * f, s[:, 1] * f
    h = np.abs(R[:, :, 0]) * ex[:, None] + np.abs(R[:, :, 1]) * ey[:, None]
    ok = op > 1 / 255.0
    lo = mu - h; hi = mu + h
    size = np.maximum(ex, ey)                                   # half-size of the quad's longer side
    hmax = h.max(1)
    key = morton(mu)
    print("P", len(mu), "valid", ok.sum(), "half-size percentiles", np.percentile(size, [5, 25, 50, 75, 95]).round(3))
    H, W = rd.shape[:2]
    rng = np.random.default_rng(0)
    TH, TW = 2, 8
    tys = rng.integers(0, H // TH, 96); txs = rng.integers(0, W // TW, 96)
    d = np.stack([rd[ty * TH:(ty + 1) * TH, tx * TW:(tx + 1) * TW].reshape(16, 3) for ty, tx in zip(tys, txs)]).astype(np.float64)
    o = ro[0, 0].astype(np.float64)

    def evaluate(name, order):
        L = tree(lo, hi, order)
        # top-down: a node is visited if its parent's box is entered and its own box is entered
        vis = None; counts = []
        for lvl in range(len(L) - 1, -1, -1):
            a, b = L[lvl]
            hit = tile_hits(a, b, o, d)
            if vis is not None:
                par = np.repeat(vis, 8, axis=1)[:, :hit.shape[1]]
                hit &= par
            vis = hit
            counts.append(hit.sum(1).mean())
        leaves = counts[-1]; nodes = sum(counts[:-1])
        print("%-34s leaf entries/tile %7.1f   node entries/tile %6.1f   (per level, root first: %s)" %
              (name, leaves, nodes, " ".join("%.0f" % c for c in counts)))
        return leaves, nodes

    base = np.argsort(np.where(ok, key, np.uint64(0x7fffffffffffffff)), kind="stable")
    evaluate("morton (current)", base)
    for ncls, edges in (("2 classes", [np.median(size)]),
                        ("3 classes", list(np.percentile(size, [33, 67]))),
                        ("4 classes", list(np.percentile(size, [25, 50, 75]))),
                        ("4 classes octave", [0.1, 0.2, 0.4]),
                        ("8 classes", list(np.percentile(size, [12.5, 25, 37.5, 50, 62.5, 75, 87.5])))):
        cls = np.searchsorted(np.asarray(edges), size).astype(np.uint64)
        k2 = (cls << np.uint64(61)) | (key >> np.uint64(2))
        evaluate("size class on top: " + ncls, np.argsort(np.where(ok, k2, np.uint64(0x7fffffffffffffff)), kind="stable"))
    # class by the box half-extent instead of the quad half-size
    for ncls, q in (("4 classes (AABB half-extent)", [25, 50, 75]),):
        cls = np.searchsorted(np.percentile(hmax, q), hmax).astype(np.uint64)
        k2 = (cls << np.uint64(61)) | (key >> np.uint64(2))
        evaluate("size class on top: " + ncls, np.argsort(np.where(ok, k2, np.uint64(0x7fffffffffffffff)), kind="stable"))


if __name__ == "__main__":
    t0 = time.time(); main(); print("%.0f s" % (time.time() - t0))
