"""Does the training step read memory it never wrote?  The C3 batch of tools/race_c3.py with the caching allocator's free memory filled with zeros,
NaN bit patterns and large finite values before each step (the step's workspace is a fresh torch.empty): every output and every gradient that is not an
atomic sum by design must come out bit-identical.  python tools/debug_uninit_train.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hybridneuralrendering_amd import scenes  # noqa: E402
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
atomic = ("aux_block", "alpha_branch", "color_final", "aux_merge_weight_block.6")

def poison(kind):
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    n = min(int(free * 0.5), 24 << 30) // 4
    t = torch.empty((n,), dtype=torch.int32, device=dev)
    t.fill_(0 if kind == "zeros" else 0x7fc00001 if kind == "nan" else 0x7f000000)
    del t
    torch.cuda.synchronize()

def step():
    out, pg, ag = train_step(path, agg, cloud.xyz, leaves[0], leaves[1], leaves[2], leaves[3], raydir, cam["campos"], cam["camrot"], cam["bg"], sc.near, sc.far,
                             cam["c2w_nearest"], cam["campos_nearest"], cam["intrinsic"], cam["images"], gt, zero_epsilon=1e-3, w_color=1.0, w_zero_one=1e-4,
                             tmid=tmid, assign_grads=False)
    cur = {("points." + k): v.clone() for k, v in pg.items()}
    cur.update({k: v.clone() for k, v in ag.items() if not k.startswith(atomic)})
    for k in ("coarse_raycolor", "decoded", "sample_pidx", "sample_loc_w", "weight", "conf_coefficient", "blend_weight", "coarse_point_opacity"):
        cur["out." + k] = out[k].clone()
    torch.cuda.synchronize()
    return cur

res = {}
for kind in ("zeros", "nan", "huge", "zeros"):
    poison(kind)
    c = step()
    if kind in res:
        bad = [k for k in c if not torch.equal(c[k], res[kind][k])]
        print("repeat %s: %d tensors differ %s" % (kind, len(bad), bad[:10]))
    else:
        res[kind] = c
for kind in ("nan", "huge"):
    bad = []
    for k in res["zeros"]:
        a, b = res["zeros"][k], res[kind][k]
        same = (a == b) | ((a != a) & (b != b)) if a.is_floating_point() else (a == b)
        if not bool(same.all()):
            bad.append((k, int((~same).sum()), int((~torch.isfinite(b)).sum()) if b.is_floating_point() else 0))
    print("free memory = %-5s: %d of %d tensors differ from the zero-filled step: %s" % (kind, len(bad), len(res["zeros"]), bad))
