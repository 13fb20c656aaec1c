"""Which tensor of the training forward differs first when repeated steps do not repeat?  The C3 batch of tools/race_c3.py in ONE fixed workspace; after
every step the forward's intermediate tensors (hnr_render_train_debug_layout) are compared with the first step's, in pipeline order.
    RACE_ITERS=600 python tools/bisect_train_forward.py       (run it beside another process that keeps the GPU busy)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hybridneuralrendering_amd import scenes, _lib  # noqa: E402
from hybridneuralrendering_amd.train import TrainPath, train_step  # noqa: E402

sys.argv = [sys.argv[0]]
args = bench.parse()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
sc, opt, agg, cloud, rnd, cam = bench.build_world(args, dev, 0)
opt.is_train = 1
path = TrainPath(rnd)
rng = np.random.default_rng(17)
x0 = int(rng.integers(args.margin, sc.w - args.margin - 56)); y0 = int(rng.integers(args.margin, sc.h - args.margin - 56))
px, py = np.meshgrid(np.arange(x0, x0 + 56), np.arange(y0, y0 + 56), indexing="ij")
pix = np.stack([px, py], axis=-1).reshape(-1, 2).astype(np.int32)
raydir = torch.from_numpy(scenes.camera_rays(pix, sc.intrinsic, sc.c2w)).to(dev)
gt = torch.rand((raydir.shape[0], 3), device=dev)
leaves = [t.clone().requires_grad_(True) for t in (cloud.emb, cloud.conf, cloud.dir, cloud.color)]
for prm in agg.parameters():
    prm.requires_grad_(True)
tmid = rnd.querier._tmid_for(float(sc.near), float(sc.far), opt.z_depth_dim, raydir.shape[0], dev)
ORDER = "fm_scratch fm vs_item row_pid Xd E Tu H1 X3 H3 H4 sigma X5 T1 T2 CF X6 vmask M1 M2 M3 X7 Y1 Y2 Y3".split()      # pipeline order
NAMES = "Xd H1 X3 H3 H4 E Tu X5 sigma T1 T2 CF X6 vmask M1 M2 M3 X7 Y1 Y2 Y3 fm row_pid vs_item fm_scratch".split()     # the library's order
def step():
    out, pg, ag = train_step(path, agg, cloud.xyz, leaves[0], leaves[1], leaves[2], leaves[3], raydir, cam["campos"], cam["camrot"], cam["bg"], sc.near, sc.far,
                             cam["c2w_nearest"], cam["campos_nearest"], cam["intrinsic"], cam["images"], gt, zero_epsilon=1e-3, w_color=1.0, w_zero_one=1e-4,
                             tmid=tmid, assign_grads=False)
    torch.cuda.synchronize()
    return out
out = step()                                                        # sizes the workspace
prm_, nbytes = path.last_step
path.workspace = torch.empty((nbytes + 256,), dtype=torch.uint8, device=dev)
lay = (ctypes.c_int64 * 64)()
n = _lib.lib().hnr_render_train_debug_layout(ctypes.byref(prm_), lay, 32)
assert n == len(NAMES), n
off0 = (-path.workspace.data_ptr()) % 256
reg = {NAMES[i]: (off0 + lay[2 * i], lay[2 * i + 1]) for i in range(n)}
def snap():
    return {k: path.workspace[o:o + b].clone() for k, (o, b) in reg.items()}
step(); step()
ref = snap(); ref_col = out["coarse_raycolor"].clone()
n_bad, first_hist = 0, {}
for it in range(int(os.environ.get("RACE_ITERS", "300"))):
    out = step()
    cur = snap()
    bad = [k for k in ORDER if not torch.equal(cur[k], ref[k])]
    if bad:
        n_bad += 1
        first_hist[bad[0]] = first_hist.get(bad[0], 0) + 1
        if n_bad <= 8:
            k = bad[0]
            a, b = cur[k].view(torch.float32) if k not in ("row_pid", "vs_item") else cur[k].view(torch.int32), ref[k].view(torch.float32) if k not in ("row_pid", "vs_item") else ref[k].view(torch.int32)
            idx = torch.nonzero(a != b).reshape(-1)
            print("step %d: differing tensors (pipeline order) %s; first = %s: %d elements, first indices %s, values %s vs %s" % (
                it, bad, k, idx.numel(), idx[:6].tolist(), a[idx[:4]].tolist(), b[idx[:4]].tolist()))
            if "fm" in bad:                                                    # per differing pixel of the feature map: which channels
                af, bf = cur["fm"].view(torch.float32).reshape(-1, 48), ref["fm"].view(torch.float32).reshape(-1, 48)
                rows_ = torch.nonzero((af != bf).any(dim=1)).reshape(-1)
                chans = sorted(set(torch.nonzero((af[rows_] != bf[rows_]).any(dim=0)).reshape(-1).tolist()))
                W_ = int(cam["images"].shape[-2]); H_ = int(cam["images"].shape[-3])
                print("    fm: %d pixels, pixel index mod 64 in %s, channels %s, (view, y, x) of the first: %s; run lengths of consecutive pixels: %s" % (
                    rows_.numel(), sorted(set((rows_ % 64).tolist()))[:20], chans, (int(rows_[0]) // (H_ * W_), (int(rows_[0]) // W_) % H_, int(rows_[0]) % W_),
                    torch.unique_consecutive(rows_ - torch.arange(rows_.numel(), device=rows_.device), return_counts=True)[1].tolist()[:12]))
print("steps with a differing forward tensor: %d of %d; first differing tensor histogram: %s" % (n_bad, int(os.environ.get("RACE_ITERS", "300")), first_hist))
