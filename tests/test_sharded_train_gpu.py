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


_WORKER_C5 = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from hybridneuralrendering_amd import scenes, parallel
from hybridneuralrendering_amd.aggregator import PointAggregator
from hybridneuralrendering_amd.render import HybridRenderer
from hybridneuralrendering_amd.train import TrainPath, render_train
from hybridneuralrendering_amd.blur import blur_update_output
from hybridneuralrendering_amd.querier import tmid_jittered
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda:0")
# BASELINE config C5 at its stated size (SURVEY 8d): scene0241-like 2 M-point cloud, random_sample='dilated' with dilation_setup 7_8_1_6
# (49 patches of 8x8 rays, strides 1..6), add_blur_sim=1 with the 12 symmetric 9x9 kernels of blur_kernel_version=2, use_frame_weight=1
sc = scenes.make_scene("scene0241", 2000000, 4)
opt = sc.opt
opt.is_train, opt.dilation_setup = 1, "7_8_1_6"
pix, pn, ps = scenes.dilated_patch_batch(sc.w, sc.h, 10, opt.dilation_setup, seed=4)
assert (pn, ps, pix.shape[0]) == (7, 8, 3136)
S = pn * ps
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rays_all = t(scenes.camera_rays(pix, sc.intrinsic, sc.c2w))
kern = t(scenes.blur_kernels_v2())[None]
assert kern.shape == (1, 12, 9, 9)
frame_weight = 0.7
g = torch.Generator().manual_seed(9)
gt = torch.rand((S * S, 3), generator=g).to(dev)
tmid = tmid_jittered(sc.near, sc.far, opt.z_depth_dim, S * S, 0.3, dev, generator=torch.Generator(device=dev).manual_seed(5))
drop = parallel.global_drop_flags(pn, ps, opt.drop_ratio).to(dev)
xyz, cam = t(sc.xyz), (t(sc.c2w[:3, 3]), t(sc.c2w[:3, :3]), t(sc.bg_color))
views = (t(sc.c2w_nearest), t(sc.c2w_nearest[:, :3, 3]), t(sc.intrinsic), t(sc.images_nearest))
torch.manual_seed(4)
agg0 = PointAggregator(opt)
with torch.no_grad():
    agg0.alpha_branch[0].weight.mul_(30.0); agg0.alpha_branch[0].bias.fill_(30.0)
sd = {k: v.clone() for k, v in agg0.state_dict().items()}

def step(ray_ids, n_total, layout, n_patches):
    agg = PointAggregator(opt); agg.load_state_dict(sd, strict=True); agg = agg.to(dev)
    leaves = [t(a).requires_grad_(True) for a in (sc.emb, sc.conf, sc.dir, sc.color)]
    path = TrainPath(HybridRenderer(opt, agg, dev))
    out = render_train(path, agg, xyz, leaves[0], leaves[1], leaves[2], leaves[3], rays_all[ray_ids], cam[0], cam[1], cam[2], sc.near, sc.far,
                       views[0], views[1], views[2], views[3], tmid=tmid[ray_ids], ray_drop=drop[ray_ids])
    col = blur_update_output(out["coarse_raycolor"][None], gt[ray_ids][None], kern, n_patches, ps, layout=layout)[0]
    m = out["ray_mask"] > 0
    loss = torch.nn.functional.mse_loss(col[m], gt[ray_ids][m]) * frame_weight * (float(m.sum()) / max(n_total, 1))
    loss.backward()
    grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in agg.parameters()]
    touched = torch.unique(out["sample_pidx"][out["sample_pidx"] >= 0]).long()
    return leaves, grads, touched, [n for n, _ in agg.named_parameters()], int(m.sum()), int(out["counts"][3])

ids, rays = parallel.shard_patches(pn, ps, world, rank)
assert ids.numel() in (24, 25)
# every ray of this closed room finds neighbours, so the global number of valid rays is S*S on both sides of the comparison
leaves, grads, touched, names, n_valid_rays, n_rows = step(rays.to(dev), S * S, "patch_major", ids.numel())
host = [x.detach().cpu() for x in grads]
parallel.allreduce_gradients(host)
emb = parallel.allreduce_point_gradients_sparse(leaves[0].grad.reshape(-1, 32).cpu(), touched.cpu())
dense = [leaves[i].grad.detach().cpu().clone() for i in (1, 2, 3)]
parallel.allreduce_gradients(dense)
if rank == 0:
    fl, fg, _, _, nv, rows = step(torch.arange(S * S, device=dev), S * S, "grid", pn)
    assert nv == S * S, nv
    worst = 0.0
    for n, a, b in zip(names, host, fg):
        b = b.cpu(); scale = float(b.abs().max())
        if scale > 0:
            worst = max(worst, float((a - b).abs().max()) / scale)
    e = float((emb - fl[0].grad.reshape(-1, 32).cpu()).abs().max() / fl[0].grad.abs().max())
    for a, i in zip(dense, (1, 2, 3)):
        e = max(e, float((a - fl[i].grad.cpu()).abs().max() / fl[i].grad.abs().max()))
    print("SHARDED_C5 rows %d weights %.2e points %.2e" % (rows, worst, e))
    assert worst < 2e-3 and e < 5e-3, (worst, e)
    print("SHARDED_C5_OK")
dist.barrier()
dist.destroy_process_group()
'''


def test_c5_full_size_patch_sharded_train_step_with_blur_module(tmp_path):
    """BASELINE config C5 ("ScanNet livingroom train step with blur-handling module, 8 x MI355X (grad path + RCCL gather)") at the
    size SURVEY 8d states -- 49 dilated 8x8 patches (dilation_setup 7_8_1_6, data/scannet_ft_dataset.py:918-949), 12 symmetric 9x9
    blur kernels (:214-242), 2 M points, frame weight -- as ONE step: two processes share the GPU, each runs forward + blur +
    backward on its 24 / 25 whole patches, gradients are summed (dense buckets + sparse point rows) and must equal the
    single-process step on all 49 patches.  Counterpart of models/base_rendering_model.py:677-745 + mvs_points_volumetric_model.py:111-148."""
    script = tmp_path / "w5.py"
    script.write_text(_WORKER_C5)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29643", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[0][-3000:] + outs[1][-3000:]
    assert "SHARDED_C5_OK" in outs[0], outs[0][-2000:]
    print([l for l in outs[0].splitlines() if l.startswith("SHARDED_C5 ")][0])
