"""Whole-frame error bound, one block: GPU colours vs the fp32 CPU oracle vs the SAME oracle evaluated in fp64 (same neighbour sets).
Tells whether a 1e-4 difference is arithmetic noise of the network (the bench's density head is scaled by 30) or a discrete mismatch."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
sys.argv = [sys.argv[0]]
args = bench.parse()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
sc, opt, agg, cloud, rnd, cam = bench.build_world(args, dev, 0)
col, out = bench.render_frame(rnd, cloud, cam, sc, 0)
gpu = col.cpu().numpy()
from oracle import query_oracle as qo, render_oracle as ro
side = 48; W = sc.w - 2 * args.margin; H = sc.h - 2 * args.margin
hp = qo.hyperparameters(sc.xyz, opt.vsize, opt.vscale, opt.kernel_size, opt.ranges, opt.radius_limit_scale)
og = qo.OracleGrid(sc.xyz, hp["origin"], hp["cell"], hp["dims"], opt.query_size, opt.P, opt.max_o)
tm = qo.tmid_table(sc.near, sc.far, opt.z_depth_dim)
sd = {k: v.detach().cpu() for k, v in agg.state_dict().items()}
for fx, fy in ((0.0, 1.0), (0.5, 0.5), (1.0, 1.0)):
    bx, by = int(fx * (W - side)), int(fy * (H - side))
    bi = ((by + np.arange(side))[:, None] * W + (bx + np.arange(side))[None, :]).reshape(-1)
    q = og.query(cam["c2w"][:3, 3], cam["rays_np"][bi], tm, opt.SR, opt.K, hp["radius2"], opt.kernel_size)
    res = {}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dt) if np.asarray(a).dtype.kind == "f" else torch.from_numpy(np.ascontiguousarray(a))
        sdd = {k: v.to(dt) if v.dtype.is_floating_point else v for k, v in sd.items()}
        qq = {k: (torch.as_tensor(v).to(dt) if torch.as_tensor(v).dtype.is_floating_point else v) if isinstance(v, (np.ndarray, torch.Tensor)) else v for k, v in q.items()} if isinstance(q, dict) else q
        torch.set_default_dtype(dt)
        with torch.no_grad():
            try:
                rb = ro.render(tt(sc.xyz), tt(sc.emb), tt(sc.conf), tt(sc.dir), tt(sc.color), sdd, qq, tt(cam["c2w"][:3, 3])[None], tt(cam["c2w"][:3, :3])[None],
                               tt(cam["rays_np"][bi])[None], tt(sc.bg_color)[None], tt(sc.c2w_nearest)[None], tt(sc.c2w_nearest[:, :3, 3])[None],
                               tt(sc.intrinsic)[None], tt(sc.images_nearest)[None], opt.vsize)["full_coarse_raycolor"][0].numpy()
                res[name] = rb.astype(np.float64)
            except Exception as e:
                print(name, "failed:", repr(e)[:300])
        torch.set_default_dtype(torch.float32)
    g = gpu[bi].astype(np.float64)
    line = "block (%d,%d): " % (bx, by)
    if "f32" in res: line += "|gpu-cpu32| %.3e  " % np.abs(g - res["f32"]).max()
    if "f64" in res: line += "|gpu-cpu64| %.3e  |cpu32-cpu64| %.3e" % (np.abs(g - res["f64"]).max(), np.abs(res["f32"] - res["f64"]).max())
    print(line)
    if "f64" in res:
        r = np.abs(g - res["f64"]).max(1).argmax(); print("  worst ray", r, "gpu", g[r], "cpu32", res["f32"][r], "cpu64", res["f64"][r])
