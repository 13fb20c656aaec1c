"""Rows per touched point of the C3 training batch (the segments hnr's per-point sums add up): histogram of the neighbour index tensor."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hybridneuralrendering_amd import scenes  # noqa: E402
from hybridneuralrendering_amd.train import TrainPath, train_step  # noqa: E402

sys.argv = [sys.argv[0]]
args = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
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
out, _, _ = train_step(path, agg, cloud.xyz, leaves[0], leaves[1], leaves[2], leaves[3], raydir, cam["campos"], cam["camrot"], cam["bg"], sc.near, sc.far,
                       cam["c2w_nearest"], cam["campos_nearest"], cam["intrinsic"], cam["images"], gt, zero_epsilon=1e-3, w_color=1.0, w_zero_one=1e-4)
p = out["sample_pidx"].cpu().numpy().reshape(-1)
p = p[p >= 0]
u, c = np.unique(p, return_counts=True)
q = np.percentile(c, [50, 90, 99, 99.9, 100])
print(json.dumps({"rows": int(p.size), "touched_points": int(u.size), "rows_per_point_p50_p90_p99_p999_max": [float(x) for x in q],
                  "points_over_64_rows": int((c > 64).sum()), "rows_in_points_over_64": int(c[c > 64].sum()),
                  "hist_le_4_8_16_32_64": [int((c <= k).sum()) for k in (4, 8, 16, 32, 64)]}))
