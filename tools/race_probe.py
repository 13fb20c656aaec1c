"""Runs the graph-free training step repeatedly on one batch and reports which gradients are not bit-identical to the first run's."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_train_gpu import _setup, _leaves
from hybridneuralrendering_amd.train import train_step
tag = sys.argv[1] if len(sys.argv) > 1 else "scannet_small_nearest0"
d, ti, opt, agg, path = _setup(tag)
near, far = d["near_far"]
tmid = torch.from_numpy(d["tmid"]).to(ti["emb"].device)
gt = torch.from_numpy(d["gt"][0]).to(ti["emb"].device)
ref = None
for it in range(int(os.environ.get("RACE_ITERS", "40"))):
    emb, conf, pdir, color = _leaves(ti)
    out, pg, ag = train_step(path, agg, ti["xyz"], emb, conf, pdir, color, ti["raydir"][0], ti["campos"][0], ti["camrotc2w"][0], ti["bg_color"][0], near, far,
                             ti["c2w_nearest"][0], ti["campos_nearest"][0], ti["intrinsic_nearest"][0], ti["images_nearest"][0], gt, zero_epsilon=float(d["zero_epsilon"]),
                             tmid=tmid, assign_grads=False)
    cur = {("pg." + k): v.clone() for k, v in pg.items()}
    cur.update({k: v.clone() for k, v in ag.items()})
    cur["col"] = out["coarse_raycolor"].clone()
    torch.cuda.synchronize()
    if ref is None: ref = cur; continue
    bad = [(k, float((cur[k] - ref[k]).abs().max()), float(ref[k].abs().max()), int((cur[k] != ref[k]).sum())) for k in ref if not torch.equal(cur[k], ref[k])]
    bad = [b for b in bad if not b[0].startswith(("aux_block", "alpha_branch", "color_final", "aux_merge_weight_block.6"))]      # (atomic sums: not bit-stable by design)
    if bad: print("iter", it, [(b[0], "%.1e" % (b[1] / (b[2] + 1e-30)), b[3]) for b in bad])
print("done")
