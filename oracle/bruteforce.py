"""Test infrastructure (never imported by the product path): a brute-force numpy restatement of the reference's raygen
loop for ONE ray -- every quad of the scene is intersected (no BVH), the hits are sorted, and the 16-candidate chunk loop
with its restart at t16 + 1e-5 is replayed (forward.cu:146-308 as restated in oracle/lrt_oracle_impl.inc; quad extent of
lib/utils/primitive_utils.py:182-224).  It shares no code with the C oracle (own intersection, no tree, float64), so the two
pin each other: tests/test_oracle_bruteforce.py; tests/tools/dense_arbiter.py uses it to arbitrate HIP-vs-oracle differences."""
import numpy as np


class QuadScene:
    def __init__(self, means, scales, rotations, opacities, scale_modifier: float = 1.0):
        self.mu = np.asarray(means, np.float64)
        s2 = np.asarray(scales, np.float64) * scale_modifier
        q = np.asarray(rotations, np.float64)
        q = q / np.linalg.norm(q, axis=1, keepdims=True)
        self.op = np.asarray(opacities, np.float64).reshape(-1)
        w, x, y, z = q.T
        R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                      2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                      2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
        self.n = R[:, :, 2]
        self.U = R[:, :, 0] / s2[:, :1]
        self.V = R[:, :, 1] / s2[:, 1:2]
        # half extent of the quad in sigma units; opacity <= 1/255: no quad (the reference builds NaN vertices there)
        self.flim = np.where(self.op > 1 / 255, np.sqrt(2 * np.log(np.maximum(255 * self.op, 1.0000001))) + 0.01, -1.0)

    def candidates(self, o, d):
        """All quads hit by the ray, sorted by t: (gidx, t, alpha)."""
        o = np.asarray(o, np.float64); d = np.asarray(d, np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            t = ((self.mu - o) * self.n).sum(1) / (self.n @ d)
        p = o + t[:, None] * d - self.mu
        u = (self.U * p).sum(1); v = (self.V * p).sum(1)
        hit = (np.abs(u) <= self.flim) & (np.abs(v) <= self.flim) & (t > 0) & np.isfinite(t)
        g = np.nonzero(hit)[0]
        g = g[np.argsort(t[g], kind="stable")]
        return g, t[g], np.minimum(0.99, self.op[g] * np.exp(-0.5 * (u[g] ** 2 + v[g] ** 2)))


def raygen_loop(g, t, alpha):
    """The reference's per-ray loop on a sorted candidate list.  Returns (composited [(gidx, t, weight)], final T,
    candidates consumed, candidates dropped by the restart epsilon [(gidx, t)])."""
    comp, drops = [], []
    T, start, i, consumed = 1.0, -1.0, 0, 0
    while True:
        while i < len(g) and not (t[i] > start):
            drops.append((int(g[i]), float(t[i]))); i += 1
        chunk = list(range(i, min(i + 16, len(g)))); i += len(chunk)
        stop = False
        for k in chunk:
            consumed += 1
            if t[k] < 0.2 or alpha[k] < 1 / 255:
                continue
            if T * (1 - alpha[k]) < 1e-4:
                stop = True
                break
            comp.append((int(g[k]), float(t[k]), float(alpha[k] * T)))
            T *= 1 - alpha[k]
        if stop or len(chunk) < 16:
            break
        start = t[chunk[-1]] + 1e-5
    return comp, T, consumed, drops


_C0 = 0.28209479177387814
_C1 = 0.4886025119029199
_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
       1.445305721320277, -0.5900435899266435)


def sh_colour(sh, direction, deg: int):
    """Colour of one Gaussian seen along `direction` (any length): real SH up to degree 3 in the 3DGS convention
    (lib/utils/sh_utils.py:58-113) + 0.5, channel 0 clamped at zero (forward.cu:67-111).  sh (M,3)."""
    d = np.asarray(direction, np.float64); x, y, z = d / np.linalg.norm(d)
    b = [_C0]
    if deg > 0:
        b += [-_C1 * y, _C1 * z, -_C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [_C2[0] * xy, _C2[1] * yz, _C2[2] * (2 * zz - xx - yy), _C2[3] * xz, _C2[4] * (xx - yy)]
    if deg > 2:
        b += [_C3[0] * y * (3 * xx - yy), _C3[1] * xy * z, _C3[2] * y * (4 * zz - xx - yy), _C3[3] * z * (2 * zz - 3 * xx - 3 * yy),
              _C3[4] * x * (4 * zz - xx - yy), _C3[5] * z * (xx - yy), _C3[6] * x * (xx - 3 * yy)]
    n = min(len(b), sh.shape[0])
    c = (np.asarray(b[:n])[:, None] * np.asarray(sh, np.float64)[:n]).sum(0) + 0.5
    c[0] = max(c[0], 0.0)
    return c
