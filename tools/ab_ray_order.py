"""A/B: the bench frame with its rays submitted in scan-line order (what the reference's chunk loop does) vs in 8x8-pixel-tile / Morton order
(one permutation of the rays before the call, one of the colours after it).  Prints per-stage times of each order, interleaved runs.
python tools/ab_ray_order.py [reps]"""
import os, sys, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

args = types.SimpleNamespace(scene="scene0241", points=2000000, width=640, height=480, margin=10, knn_order=None)
dev = torch.device("cuda:0")
sc, opt, agg, cloud, rnd, cam = bench.build_world(args, dev, 0)
pix = cam["pix"]
W = sc.w - 2 * args.margin


def order(kind):
    x, y = pix[:, 0].astype(np.int64) - args.margin, pix[:, 1].astype(np.int64) - args.margin
    if kind == "scan":
        return np.arange(pix.shape[0])
    if kind.startswith("tile"):
        ts = int(kind[4:])
        key = ((y // ts) * ((W + ts - 1) // ts) + (x // ts)) * (ts * ts) + (y % ts) * ts + (x % ts)
        return np.argsort(key, kind="stable")
    if kind == "morton":
        def spread(v):
            v = v & 0xFFFF
            v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F; v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555
            return v
        return np.argsort(spread(x) | (spread(y) << 1), kind="stable")
    raise ValueError(kind)


kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ["scan", "tile8", "tile16", "morton"]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
perm = {k: torch.from_numpy(order(k)).to(dev) for k in kinds}
rays = {k: cam["raydir"][perm[k]].contiguous() for k in kinds}
ref = None
res = {k: [] for k in kinds}
for rep in range(reps + 1):
    for k in kinds:
        c2 = dict(cam, raydir=rays[k])
        timers = {}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        col, out = bench.render_frame(rnd, cloud, c2, sc, 0, timers=timers)
        full = torch.empty_like(col); full[perm[k]] = col
        e1.record(); torch.cuda.synchronize()
        if ref is None: ref = full.clone()
        if rep == 0:
            print(k, "max |d colour| vs scan order: %.2e" % float((full - ref).abs().max()), "counts", out["counts"].cpu().numpy()[[1, 3, 6]])
            continue
        st = timers["_stage_events"][0].elapsed_ms()
        res[k].append((e0.elapsed_time(e1), st))
for k in kinds:
    t = np.array([r[0] for r in res[k]])
    keys = list(res[k][0][1].keys())
    st = {n: float(np.mean([r[1][n] for r in res[k]])) for n in keys}
    print("%-7s frame ms: mean %.3f min %.3f | " % (k, t.mean(), t.min()) + " ".join("%s %.3f" % (n, st[n]) for n in keys))
