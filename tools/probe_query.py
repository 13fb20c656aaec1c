"""Perf probe of the query stage on the headline config (not part of the product)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybridneuralrendering_amd import scenes, querier as Q
from hybridneuralrendering_amd._lib import CNT

name = sys.argv[1] if len(sys.argv) > 1 else "scene0241"
PAD = os.environ.get("PROBE_PAD", "1") != "0"                       # 0: un-padded outputs, what bench.py's roofline_query times
ORDER = int(os.environ.get("PROBE_KNN_ORDER", "0"))              # 1: hnr_query_params.knn_order = 1 (sorted neighbour lists)
N = int(float(sys.argv[2])) if len(sys.argv) > 2 else 2000000
dev = torch.device("cuda:0")
t0 = time.time()
sc = scenes.make_scene(name, N, 2)
print("scene gen %.1fs" % (time.time() - t0))
opt = sc.opt
margin = 10 if name.startswith("scene") else 0
pix = scenes.pixel_grid(sc.w, sc.h, margin)
rays = torch.from_numpy(scenes.camera_rays(pix, sc.intrinsic, sc.c2w)).to(dev)
xyz = torch.from_numpy(sc.xyz).to(dev)
mn, mx = Q.points_bounds(xyz)
rl, ranges_np, cell, dims, _ = Q.compute_hyperparameters(mn, mx, opt.vsize, opt.vscale, opt.kernel_size, opt.ranges, opt.radius_limit_scale)
print("dims", dims, "cell", cell, "R", rays.shape[0])
torch.cuda.synchronize(); t0 = time.time()
g = Q.VoxelGrid(xyz, ranges_np[:3], cell, dims, opt.query_size, opt.P, opt.max_o)
torch.cuda.synchronize(); print("grid build %.1f ms" % ((time.time() - t0) * 1e3), g.stats)
campos = torch.from_numpy(sc.c2w[:3, 3].copy()).to(dev)
tm = Q.tmid_table(sc.near, sc.far, opt.z_depth_dim, device=dev)
r2 = np.float32(rl ** 2)
for it in range(3):
    res = Q.march_query(g, campos, rays, tm, opt.SR, opt.K, r2, opt.kernel_size, knn_order=ORDER, pad=PAD)
torch.cuda.synchronize()
c = res["counts"].cpu().numpy()
print({k: int(c[v]) for k, v in CNT.items()})
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 10
for it in range(n):
    res = Q.march_query(g, campos, rays, tm, opt.SR, opt.K, r2, opt.kernel_size, knn_order=ORDER, pad=PAD)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
R = rays.shape[0]
s, v, cand = c[CNT["SAMPLES"]], c[CNT["CELLS_VISITED"]], c[CNT["CANDIDATES"]]
alg = R * (12 + (opt.z_depth_dim + 7) // 8 + 1) + s * (12 + 27 * 4 + 4 * opt.K) + 4 * v + 16 * cand
print("march+knn %.3f ms  -> %.1f Mrays/s; algorithmic %.1f MB -> %.1f GB/s (%.1f%% of 8 TB/s); samples/ray %.2f cells/sample %.2f cand/sample %.1f" % (
    ms, R / ms / 1e3, alg / 1e6, alg / ms / 1e6, alg / ms / 1e6 / 8000 * 100, s / R, v / max(s, 1), cand / max(s, 1)))
