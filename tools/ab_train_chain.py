"""Per-tensor checksums of one training step (outputs, point gradients, weight gradients) -- run once with HNR_TRAIN_CHAIN_WS=1 and once with 0 and
diff the output: which tensors depend on the form of the training chain kernel.  python tools/ab_train_chain.py"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_train_gpu import _setup, _leaves
from hybridneuralrendering_amd.train import train_step
for tag in ("scannet_small",):
    d, ti, opt, agg, path = _setup(tag)
    near, far = d["near_far"]
    tmid = torch.from_numpy(d["tmid"]).to(ti["emb"].device)
    gt = torch.from_numpy(d["gt"][0]).to(ti["emb"].device)
    emb, conf, pdir, color = _leaves(ti)
    out, pg, ag = train_step(path, agg, ti["xyz"], emb, conf, pdir, color, ti["raydir"][0], ti["campos"][0], ti["camrotc2w"][0], ti["bg_color"][0], near, far,
                             ti["c2w_nearest"][0], ti["campos_nearest"][0], ti["intrinsic_nearest"][0], ti["images_nearest"][0], gt,
                             zero_epsilon=float(d["zero_epsilon"]), tmid=tmid, assign_grads=False)
    sha = lambda t: hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()[:12]
    for k in sorted(out):
        if torch.is_tensor(out[k]): print(tag, "out", k, sha(out[k]), float(out[k].double().abs().sum()))
    for k in sorted(pg): print(tag, "pg", k, sha(pg[k]), float(pg[k].double().abs().sum()))
    for k in sorted(ag): print(tag, "ag", k, sha(ag[k]), float(ag[k].double().abs().sum()))
    if len(sys.argv) > 1:
        import numpy as np
        arrs = {"out." + k: out[k].detach().cpu().numpy() for k in out if torch.is_tensor(out[k])}
        arrs.update({"pg." + k: pg[k].detach().cpu().numpy() for k in pg}); arrs.update({"ag." + k: ag[k].detach().cpu().numpy() for k in ag})
        np.savez(sys.argv[1], **arrs)
