"""Randomised parity sweep of the full HIP path against the CPU oracle: random cameras, random weights, forward colours and
(every 4th case) the gradients of a training step.  Not a test; run occasionally:  python tools/stress_render.py [n_cases]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridneuralrendering_amd import scenes  # noqa: E402
from hybridneuralrendering_amd.aggregator import PointAggregator  # noqa: E402
from hybridneuralrendering_amd.render import HybridRenderer, PointCloud  # noqa: E402
from hybridneuralrendering_amd.train import TrainPath, render_train  # noqa: E402
from oracle import query_oracle as qo, render_oracle as ro  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0")
rng = np.random.default_rng(7)
worst_fwd, worst_grad, bad = 0.0, 0.0, 0
for i in range(n_cases):
    name = "scene0241" if i % 3 else "lego"
    sc = scenes.make_scene(name, int(rng.choice([6000, 15000])), int(rng.integers(1, 10 ** 6)), w=48, h=40,
                           size=(1.0, 0.8, 0.6) if name == "scene0241" else None)
    opt = sc.opt
    opt.SR = int(rng.choice([8, 24, 40]))
    torch.manual_seed(int(rng.integers(1, 10 ** 6)))
    agg = PointAggregator(opt)
    with torch.no_grad():
        agg.alpha_branch[0].weight.mul_(float(rng.uniform(5, 40)))
        agg.alpha_branch[0].bias.fill_(float(rng.uniform(0, 40)))
    sd = {k: v.detach().clone() for k, v in agg.state_dict().items()}
    agg = agg.to(dev)
    pix = scenes.pixel_grid(sc.w, sc.h, 2)
    sel = np.sort(rng.choice(pix.shape[0], size=400, replace=False))
    rays = scenes.camera_rays(pix[sel], sc.intrinsic, sc.c2w)
    hp = qo.hyperparameters(sc.xyz, opt.vsize, opt.vscale, opt.kernel_size, opt.ranges, opt.radius_limit_scale)
    og = qo.OracleGrid(sc.xyz, hp["origin"], hp["cell"], hp["dims"], opt.query_size, opt.P, opt.max_o)
    train = i % 4 == 0
    tm = qo.tmid_table(sc.near, sc.far, opt.z_depth_dim)
    if train:
        tm = (tm[None].repeat(400, 0) * (1 + 2e-3 * rng.uniform(-1, 1, size=(400, tm.shape[0])))).astype(np.float32)
    q = og.query(sc.c2w[:3, 3], rays, tm, opt.SR, opt.K, hp["radius2"], opt.kernel_size)
    c = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    t = lambda a: c(a).to(dev)
    args_cpu = (c(sc.c2w[:3, 3])[None], c(sc.c2w[:3, :3])[None], c(rays)[None], c(sc.bg_color)[None], c(sc.c2w_nearest)[None],
                c(sc.c2w_nearest[:, :3, 3])[None], c(sc.intrinsic)[None], c(sc.images_nearest)[None], opt.vsize)
    rnd = HybridRenderer(opt, agg, dev)
    if not train:
        with torch.no_grad():
            ref = ro.render(c(sc.xyz), c(sc.emb), c(sc.conf), c(sc.dir), c(sc.color), sd, q, *args_cpu)
        out = rnd.render_rays(PointCloud(t(sc.xyz), t(sc.emb), t(sc.conf), t(sc.dir), t(sc.color)), t(rays), t(sc.c2w[:3, 3]), t(sc.c2w[:3, :3]),
                              t(sc.bg_color), sc.near, sc.far, t(sc.c2w_nearest), t(sc.c2w_nearest[:, :3, 3]), t(sc.intrinsic), t(sc.images_nearest),
                              w2c_nearest=torch.inverse(c(sc.c2w_nearest)).to(dev))
        e = float((out["coarse_raycolor"].cpu() - ref["full_coarse_raycolor"][0]).abs().max())
        worst_fwd = max(worst_fwd, e)
        ok = e < 3e-4
    else:
        opt.is_train, opt.dilation_setup = 1, "5_4_1_8"
        gt = rng.uniform(0, 1, size=(1, 400, 3)).astype(np.float32)
        _, losses, gref = ro.train_step(c(sc.xyz), c(sc.emb), c(sc.conf), c(sc.dir), c(sc.color), sd, q, *args_cpu[:8], opt.vsize, c(gt), 1e-3,
                                        ro.drop_patch_rays(4, 5, opt.drop_ratio))
        leaves = [t(a).requires_grad_(True) for a in (sc.emb, sc.conf, sc.dir, sc.color)]
        for prm in agg.parameters():
            prm.requires_grad_(True)
        o = render_train(TrainPath(rnd), agg, t(sc.xyz), leaves[0], leaves[1], leaves[2], leaves[3], t(rays), t(sc.c2w[:3, 3]), t(sc.c2w[:3, :3]),
                         t(sc.bg_color), sc.near, sc.far, t(sc.c2w_nearest), t(sc.c2w_nearest[:, :3, 3]), t(sc.intrinsic), t(sc.images_nearest), tmid=t(tm))
        m = o["ray_mask"] > 0
        val = torch.clamp(o["conf_coefficient"][m], 1e-3, 1 - 1e-3)
        loss = torch.nn.functional.mse_loss(o["coarse_raycolor"][m], t(gt[0])[m]) + 1e-4 * torch.mean(torch.log(val) + torch.log(1 - val))
        loss.backward()
        e = abs(loss.item() - losses[0]) / max(abs(losses[0]), 1e-6)
        got = {"neural_points.points_embeding": leaves[0].grad, "neural_points.points_conf": leaves[1].grad, "neural_points.points_dir": leaves[2].grad,
               "neural_points.points_color": leaves[3].grad}
        got.update({"aggregator." + k: v.grad for k, v in agg.named_parameters() if v.grad is not None})
        ge, worst_key = 0.0, ""
        for k, r in gref.items():
            r = r.numpy()
            if r.size > 1 and np.abs(r).max() > 0:
                d = np.abs(got[k].detach().cpu().numpy().reshape(r.shape) - r) / np.abs(r).max()
                if float(d.max()) > ge:
                    ge, worst_key = float(d.max()), "%s (%d of %d entries above 1e-3 of max)" % (k, int((d > 1e-3).sum()), d.size)
        worst_grad = max(worst_grad, ge)
        ok = e < 1e-4 and ge < 5e-3
        if not ok and ge >= 5e-3:
            # the yardstick: the same graph in fp64 -- how far is the ORACLE's own fp32 result from it on the entries where the GPU differs?  (A pre-activation
            # within rounding of a LeakyReLU kink, or a sample within an ulp of a pixel border, flips a discrete choice between two fp32 evaluations.)
            _, _, g64 = ro.train_step(c(sc.xyz), c(sc.emb), c(sc.conf), c(sc.dir), c(sc.color), sd, q, *args_cpu[:8], opt.vsize, c(gt), 1e-3,
                                      ro.drop_patch_rays(4, 5, opt.drop_ratio), dtype=torch.float64)
            k = worst_key.split(" ")[0]
            r32, r64, gg = gref[k].numpy().astype(np.float64), g64[k].numpy().astype(np.float64), got[k].detach().cpu().numpy().astype(np.float64).reshape(gref[k].shape)
            sc_ = np.abs(r64).max()
            print("    fp64 yardstick on %s: |gpu - fp64| %.1e, |oracle fp32 - fp64| %.1e, |gpu - oracle fp32| %.1e (all / max|fp64|)" % (
                k, np.abs(gg - r64).max() / sc_, np.abs(r32 - r64).max() / sc_, np.abs(gg - r32).max() / sc_))
            if np.abs(gg - r64).max() <= max(2.0 * np.abs(r32 - r64).max(), 1e-3 * sc_):
                ok = True; bad_note = "    -> within the oracle's own fp32-vs-fp64 distance: counted as ok"
                print(bad_note)
        opt.is_train = 0
    if not ok:
        bad += 1
        if train:
            print("    worst gradient: " + worst_key)
    print("case %2d %-9s SR=%2d valid rays %3d  %s  -> %s" % (i, name, opt.SR, int(q["ray_mask"].sum()),
                                                            ("loss rel %.1e grad %.1e" % (e, ge)) if train else ("max|d colour| %.1e" % e), "ok" if ok else "MISMATCH"))
print("stress_render: %d cases, %d mismatches; worst forward %.1e, worst gradient %.1e of max" % (n_cases, bad, worst_fwd, worst_grad))
sys.exit(1 if bad else 0)
