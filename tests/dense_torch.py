"""Dense (all-pairs) PyTorch restatement of the reference FORWARD
(DLT/optix_tracer/forward.cu:146-308), differentiable by autograd.
TEST INFRASTRUCTURE: used to check the oracle's analytic backward.

Non-differentiable decisions (hit/miss, alpha<1/255 skip, T<1e-4 stop,
0.99 clamp, channel-0 clamp) are evaluated on detached values exactly like
the kernel does; the 16-hit chunk restart epsilon is not modelled (it only
drops hits that lie within 1e-5 of a chunk boundary).

``bg_factor=2`` reproduces the reference backward's double-counted
background term: the reference gradient equals
autograd of this forward with the background counted twice.
"""
import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def sh_basis(deg, d):
    d = d / d.norm(dim=-1, keepdim=True)
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    b = [torch.full_like(x, SH_C0)]
    if deg > 0:
        b += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2 * zz - xx - yy), SH_C2[3] * xz, SH_C2[4] * (xx - yy)]
    if deg > 2:
        b += [SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * xy * z, SH_C3[2] * y * (4 * zz - xx - yy),
              SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy), SH_C3[4] * x * (4 * zz - xx - yy),
              SH_C3[5] * z * (xx - yy), SH_C3[6] * x * (xx - 3 * yy)]
    return torch.stack(b, -1)  # (..., (deg+1)^2)


def rotmat(q):
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    return torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
        torch.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
        torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)], -2)


def render(ray_o, ray_d, means, scales, quats, opac, shs, deg, bg, bg_factor=1.0):
    """ray_o, ray_d: (N,3); means (P,3); scales (P,2); quats (P,4) (w,x,y,z);
    opac (P,); shs (P,M,3); returns out (N,9)."""
    N, P = ray_o.shape[0], means.shape[0]
    R = rotmat(quats)                       # (P,3,3)
    n = R[:, :, 2]
    a = R[:, :, 0] / scales[:, 0:1]
    b = R[:, :, 1] / scales[:, 1:2]
    co = means[None] - ray_o[:, None]       # (N,P,3)
    num = (co * n[None]).sum(-1)
    den = (ray_d[:, None] * n[None]).sum(-1)
    t = num / den                           # (N,P)
    x = ray_o[:, None] + t[..., None] * ray_d[:, None]
    pm = x - means[None]
    u = (pm * a[None]).sum(-1)
    v = (pm * b[None]).sum(-1)
    f = torch.sqrt(2 * torch.log(255 * opac)) + 0.01
    with torch.no_grad():
        hit = (u.abs() <= f[None]) & (v.abs() <= f[None]) & (t >= 0.2) & torch.isfinite(t)
        tkey = torch.where(hit, t, torch.full_like(t, float("inf")))
        order = torch.argsort(tkey, dim=1)
        nhit = hit.sum(1)
    kmax = int(nhit.max().item()) if N > 0 else 0
    basis = sh_basis(deg, ray_d)            # (N,nb)
    nb = basis.shape[-1]
    C = torch.zeros(N, 3, dtype=means.dtype)
    D = torch.zeros(N, dtype=means.dtype)
    Wt = torch.zeros(N, dtype=means.dtype)
    T = torch.ones(N, dtype=means.dtype)
    alive = torch.ones(N, dtype=torch.bool)
    rows = torch.arange(N)
    for k in range(kmax):
        g = order[:, k]
        valid = alive & (k < nhit)
        uu, vv, tt = u[rows, g], v[rows, g], t[rows, g]
        G = torch.exp(-0.5 * (uu * uu + vv * vv))
        al = torch.minimum(opac[g] * G, torch.full_like(G, 0.99))
        with torch.no_grad():
            contrib = valid & (al >= 1.0 / 255.0)
            testT = T * (1 - al)
            stop = contrib & (testT < 1e-4)
            alive = alive & ~stop
            contrib = contrib & ~stop
        w = al * T
        col = (basis[:, :, None] * shs[g][:, :nb, :]).sum(1) + 0.5
        col = torch.cat([col[:, 0:1].clamp_min(0.0), col[:, 1:]], 1)
        m = contrib.to(means.dtype)
        C = C + (m * w)[:, None] * col
        D = D + m * w * tt
        Wt = Wt + m * w
        T = torch.where(contrib, T * (1 - al), T)
    out = torch.zeros(N, 9, dtype=means.dtype)
    out = torch.cat([C + bg_factor * T[:, None] * bg[None], D[:, None], Wt[:, None],
                     torch.zeros(N, 3, dtype=means.dtype), T[:, None]], 1)
    return out
