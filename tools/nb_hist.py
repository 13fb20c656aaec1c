import sys, os, types
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
args = types.SimpleNamespace(scene="scene0241", points=2e6, width=640, height=480, margin=10)
dev = torch.device("cuda:0")
sc, opt, agg, cloud, rnd, cam = bench.build_world(args, dev, 0)
col, out = bench.render_frame(rnd, cloud, cam, sc, 0)
torch.cuda.synchronize()
p = out["sample_pidx"]; ns = out["ray_nsamp"]
SR = p.shape[1]
kept = torch.arange(SR, device=dev)[None, :] < ns[:, None].long()
cnt = (p >= 0).sum(dim=-1)[kept]
h = torch.bincount(cnt, minlength=9).cpu().numpy()
print("neighbour-count histogram of kept samples (0..8):", h, "fractions", np.round(h / h.sum(), 4))
v = h[1:]
print("valid samples %d, rows if padded to 8: %d, neighbours %d, padded to 4|8: %d" % (v.sum(), 8 * v.sum(), (np.arange(1, 9) * v).sum(), 4 * v[:4].sum() + 8 * v[4:].sum()))
