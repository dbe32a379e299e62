"""Developer probe of --deterministic: where do an uninterrupted run and a run restored from a checkpoint part?  One process: K iterations, checkpoint,
one more iteration with its gradients kept; then a fresh scene + fresh tracer restored from the checkpoint, the same iteration; compares losses, gradients, parameters."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import make_sequence
from lidar_rt_amd import renderer, sequence, training, train

dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp()
data = os.path.join(tmp, "seq")
make_sequence.make("kitti360_dynamic", data, n_frames=4, scale=0.1)
renderer.deferred_accum = True; renderer.deterministic = os.environ.get("DET", "1") == "1"
seq = sequence.load_sequence(data, dev)
opt = training.default_options(); opt.iterations = 24
bg = torch.tensor([0.0, 0.0, 1.0], device=dev)
K = int(os.environ.get("K", "12"))


def fresh():
    renderer.tracer_2dgs = None
    torch.manual_seed(0)
    sc = sequence.scene_from_sequence(seq, max_points=60000, seed=0)
    sc.training_setup(opt)
    return sc


_orig_rt = renderer.raytracing
UP = {}
def _rt(*a, **k):
    r = _orig_rt(*a, **k)
    for name in ("depth", "intensity", "raydrop"):
        if r[name].requires_grad:
            UP["fwd_" + name] = r[name].detach().clone()
            r[name].register_hook(lambda g, n=name: UP.__setitem__("dL_" + n, g.detach().clone()))
    return r
renderer.raytracing = _rt



def one(scene, it, keep=False):
    torch.manual_seed(it)
    frame = train.frame_of(0, it, seq.train_frames)
    grads = {}
    if keep:      # the gradients as the optimizer sees them: hook every parameter
        for ai, g in enumerate(scene.gaussians_assets):
            for name in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest"):
                pr = getattr(g, name, None)
                if pr is not None and pr.requires_grad:
                    pr.register_hook(lambda gr, k=(ai, name): grads.__setitem__(k, gr.detach().clone()))
    res = training.training_step(scene, seq.frames, frame, it, opt, bg, dynamic=True)
    return float(res["loss"]), grads


a = fresh()
for it in range(1, K + 1):
    one(a, it)
ck = os.path.join(tmp, "c.pth"); a.save(K, ck)
la, ga = one(a, K + 1, keep=True)
pa = [g.capture() for g in a.gaussians_assets]
ua = dict(UP); UP.clear()
b = fresh()
b.restore(torch.load(ck, map_location=dev, weights_only=False)[0], opt)
lb, gb = one(b, K + 1, keep=True)
pb = [g.capture() for g in b.gaussians_assets]
ub = dict(UP); UP.clear()
c = fresh()
c.restore(torch.load(ck, map_location=dev, weights_only=False)[0], opt)
lc, gc = one(c, K + 1, keep=True)
uc = dict(UP)
for k in sorted(ua):
    print("boundary", k, "uninterrupted vs restored:", "identical" if torch.equal(ua[k], ub[k]) else "DIFFER %d" % int((ua[k] != ub[k]).sum()),
          "| restored vs restored again:", "identical" if torch.equal(ub[k], uc[k]) else "DIFFER %d" % int((ub[k] != uc[k]).sum()))
st_ = renderer.tracer_2dgs.optix_context
print("tracer:", {k: st_.get_option(k, dev) for k in ("hit_cap", "near_rays_last", "last_bwd_speculative", "deterministic", "deferred_accum", "carry_order")})
import ctypes as C
idx, h = st_.handle(dev); hn = __import__("numpy").empty(66 * 1030, "int32"); st_._lib.lrt_debug_read.restype = C.c_longlong
st_._lib.lrt_debug_read(h, 5, hn.ctypes.data_as(C.c_void_p), C.c_longlong(hn.nbytes), None)
print("composited hits per ray: max", int(hn.max()), "mean %.1f" % float(hn.mean()))
d0 = (ub["fwd_intensity"] != uc["fwd_intensity"]).reshape(66, 1030)
ys, xs = d0.nonzero(as_tuple=True)
print("differing pixels by row:", torch.bincount(ys, minlength=66).tolist())
print("their hit counts:", hn.reshape(66, 1030)[ys.cpu().numpy(), xs.cpu().numpy()][:40].tolist())
print("max abs diff", float((ub["fwd_intensity"] - uc["fwd_intensity"]).abs().max()))
print("gradients restored vs restored again:", {k: ("identical" if torch.equal(gb[k], gc[k]) else "DIFFER") for k in sorted(gb) if k[0] < 2})
print("loss at", K + 1, la, lb, "equal" if la == lb else "DIFFER")
for k in sorted(ga):
    x, y = ga[k], gb[k]
    print("grad", k, "identical" if torch.equal(x, y) else "DIFFER %d of %d, max abs %.3e (max |g| %.3e)" % (int((x != y).sum()), x.numel(), float((x - y).abs().max()), float(x.abs().max())))
for ai, (x, y) in enumerate(zip(pa, pb)):
    for i in (1, 2, 3, 4, 5, 6):
        if not torch.equal(x[i], y[i]):
            print("param asset", ai, "field", i, "DIFFER", int((x[i] != y[i]).sum()), "of", x[i].numel())
print("done")
