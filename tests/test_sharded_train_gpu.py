"""Config C5 control flow on ONE GPU: two processes share the device and exchange gradients over gloo (host copies).  A training
batch sharded by whole patches (parallel.shard_patches), local losses scaled by n_local / n_total, gradients summed with
parallel.allreduce_gradients / allreduce_point_gradients_sparse must equal the single-process step on the full batch."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from tests.golden_io import load_train, torch_inputs
from hybridneuralrendering_amd import scenes, parallel
from hybridneuralrendering_amd.aggregator import PointAggregator
from hybridneuralrendering_amd.render import HybridRenderer
from hybridneuralrendering_amd.train import TrainPath, render_train
from hybridneuralrendering_amd.blur import blur_update_output
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
d = load_train("scannet_small")
dev = torch.device("cuda:0")
opt = scenes.default_opt(**d["opt"])
pn, ps = 7, 4
ti = torch_inputs(d, dev)
near, far = d["near_far"]
tmid = torch.from_numpy(d["tmid"]).to(dev)
gt = torch.from_numpy(d["gt"][0]).to(dev)
drop = parallel.global_drop_flags(pn, ps, opt.drop_ratio).to(dev)
g = torch.Generator().manual_seed(5)
kern = torch.rand((6, 5, 5), generator=g) ** 3
kern = (kern / kern.sum(dim=(1, 2), keepdim=True)).to(dev)[None]

def step(ray_ids, n_total, layout, n_patches):
    agg = PointAggregator(opt); agg.load_state_dict(d["sd"], strict=True); agg = agg.to(dev)
    leaves = [ti[k].clone().requires_grad_(True) for k in ("emb", "conf", "pdir", "color")]
    path = TrainPath(HybridRenderer(opt, agg, dev))
    out = render_train(path, agg, ti["xyz"], leaves[0], leaves[1], leaves[2], leaves[3], ti["raydir"][0][ray_ids], ti["campos"][0], ti["camrotc2w"][0],
                       ti["bg_color"][0], near, far, ti["c2w_nearest"][0], ti["campos_nearest"][0], ti["intrinsic_nearest"][0], ti["images_nearest"][0],
                       tmid=tmid[ray_ids], ray_drop=drop[ray_ids])
    col = blur_update_output(out["coarse_raycolor"][None], gt[ray_ids][None], kern, n_patches, ps, layout=layout)[0]
    loss = torch.nn.functional.mse_loss(col, gt[ray_ids]) * parallel.loss_scale(ray_ids.numel(), n_total)     # mean over rays -> global mean
    loss.backward()
    grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in agg.parameters()]
    touched = torch.unique(out["sample_pidx"][out["sample_pidx"] >= 0]).long()
    return leaves, grads, touched, [n for n, _ in agg.named_parameters()]

S = pn * ps
ids, rays = parallel.shard_patches(pn, ps, world, rank)
leaves, grads, touched, names = step(rays.to(dev), S * S, "patch_major", ids.numel())
# sum over ranks (gloo: host copies)
host = [t.detach().cpu() for t in grads]
parallel.allreduce_gradients(host)
emb = parallel.allreduce_point_gradients_sparse(leaves[0].grad[0].cpu(), touched.cpu())
dense = [leaves[i].grad.detach().cpu().clone() for i in (1, 2, 3)]
parallel.allreduce_gradients(dense)
if rank == 0:
    # the same batch in one process, grid layout
    fl, fg, _, _ = step(torch.arange(S * S, device=dev), S * S, "grid", pn)
    worst = 0.0
    for n, a, b in zip(names, host, fg):
        b = b.cpu(); sc = float(b.abs().max())
        if sc > 0:
            worst = max(worst, float((a - b).abs().max()) / sc)
    e = float((emb - fl[0].grad[0].cpu()).abs().max() / fl[0].grad.abs().max())
    for a, i in zip(dense, (1, 2, 3)):
        e = max(e, float((a - fl[i].grad.cpu()).abs().max() / fl[i].grad.abs().max()))
    print("SHARDED_TRAIN weights %.2e points %.2e" % (worst, e))
    assert worst < 1e-3 and e < 3e-3, (worst, e)
    print("SHARDED_TRAIN_OK")
dist.barrier()
dist.destroy_process_group()
'''


def test_patch_sharded_train_step_equals_single_process(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[0][-3000:] + outs[1][-3000:]
    assert "SHARDED_TRAIN_OK" in outs[0], outs[0][-2000:]
