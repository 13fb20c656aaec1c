"""Config C5 control flow on ONE GPU: two processes share the device and exchange gradients over gloo (host copies).  A training
batch sharded by whole patches (parallel.shard_patches) must give the gradients of the single-process step on the full batch:
 * the small fixture through the autograd form + the legacy bucketed / sparse all-reduces (parallel.allreduce_gradients, allreduce_point_gradients_sparse);
 * BASELINE config C5 at full size through the PRODUCTION path (train.train_step + parallel.allreduce_weight_grads + parallel.PointGradExchange)."""
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
from hybridneuralrendering_amd.train import TrainPath, train_step
from hybridneuralrendering_amd.querier import tmid_jittered
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ.get("HNR_TEST_BACKEND", "gloo")          # gloo: both ranks share cuda:0, collectives on host copies; nccl: one GPU per rank, RCCL
dist.init_process_group(backend, rank=rank, world_size=world)
dev = torch.device("cuda:%d" % (rank if backend == "nccl" else 0))
torch.cuda.set_device(dev)
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
tmid = tmid_jittered(sc.near, sc.far, opt.z_depth_dim, S * S, 0.3, "cpu", generator=torch.Generator().manual_seed(5)).to(dev)
drop = parallel.global_drop_flags(pn, ps, opt.drop_ratio).to(dev)
xyz, cam = t(sc.xyz), (t(sc.c2w[:3, 3]), t(sc.c2w[:3, :3]), t(sc.bg_color))
views = (t(sc.c2w_nearest), t(sc.c2w_nearest[:, :3, 3]), t(sc.intrinsic), t(sc.images_nearest))
torch.manual_seed(4)
agg0 = PointAggregator(opt)
with torch.no_grad():
    agg0.alpha_branch[0].weight.mul_(30.0); agg0.alpha_branch[0].bias.fill_(30.0)
sd = {k: v.clone() for k, v in agg0.state_dict().items()}
KEYS = ("points_embeding", "points_conf", "points_dir", "points_color")

def step(ray_ids, layout, n_patches):
    # the production step bench.py times: train.train_step (forward -> hnr_blur_select -> loss kernels -> blur backward -> backward, no autograd graph)
    agg = PointAggregator(opt); agg.load_state_dict(sd, strict=True); agg = agg.to(dev)
    leaves = [t(a).requires_grad_(True) for a in (sc.emb, sc.conf, sc.dir, sc.color)]
    path = TrainPath(HybridRenderer(opt, agg, dev))
    out, pg, ag = train_step(path, agg, xyz, leaves[0], leaves[1], leaves[2], leaves[3], rays_all[ray_ids].contiguous(), cam[0], cam[1], cam[2], sc.near, sc.far,
                             views[0], views[1], views[2], views[3], gt[ray_ids].contiguous(), zero_epsilon=1e-3, w_color=1.0, w_zero_one=1e-4,
                             frame_weight=frame_weight, tmid=tmid[ray_ids].contiguous(), ray_drop=drop[ray_ids].contiguous(), assign_grads=False,
                             blur_kernels=kern, patch_num=n_patches, patch_size=ps, patch_layout=layout)
    torch.cuda.synchronize()
    TrainPath.check_status(out)
    return out, pg, ag

ids, rays = parallel.shard_patches(pn, ps, world, rank)
assert ids.numel() in (24, 25)
out, pg, ag = step(rays.to(dev), "patch_major", int(ids.numel()))
Sv = out["_saved"]
nv = out["loss"][3:4]
host = (lambda x: x.cpu()) if backend == "gloo" else (lambda x: x)
# collective 1: ONE all-reduce of the flat weight-gradient buffer carrying the ranks' valid-ray counts
offs = {n: ((a.data_ptr() - Sv.flat.data_ptr()) // 4, a.shape) for n, a in ag.items()}
flat = host(Sv.flat)
parallel.allreduce_weight_grads(flat, host(nv), Sv.flat_payload)
# collective 2: ONE fixed-capacity all-gather of packed (point id | 39 floats) records; the capacity is agreed once
tids, tcnt = TrainPath.touched_points(Sv)
cap = torch.tensor([int(tcnt.item())], dtype=torch.int64, device="cpu" if backend == "gloo" else dev)
dist.all_reduce(cap, op=dist.ReduceOp.MAX)
ex = parallel.PointGradExchange(int(cap.item()) + 64)
bufs = [host(pg[k]) for k in KEYS]
rec = ex.pack(bufs, host(tids), host(tcnt), host(nv))
tot, over = ex.apply(ex.exchange(rec), bufs, rank)
ex.raise_on_overflow(over)
assert int(float(tot)) == S * S                                   # every ray of this closed room finds neighbours
if rank == 0:
    o1, pg1, ag1 = step(torch.arange(S * S, device=dev), "grid", pn)
    assert int(o1["loss"][3].item()) == S * S
    worst, wname = 0.0, None
    for n, (o, shp) in offs.items():
        a = flat[o:o + int(np.prod(shp))].reshape(shp).cpu(); b = ag1[n].cpu(); scale = float(b.abs().max())
        if scale > 0:
            e = float((a - b).abs().max()) / scale
            if e > worst: worst, wname = e, n
    e, ename = 0.0, None
    for k, a in zip(KEYS, bufs):
        b = pg1[k].cpu(); x = float((a.cpu().reshape(b.shape) - b).abs().max() / b.abs().max())
        if x > e: e, ename = x, k
    # the loss of the whole batch = the valid-ray-weighted mean of the ranks' colour terms (checked through the gradients above); its value on rank 0's share is finite
    assert bool(torch.isfinite(out["loss"]).all())
    print("SHARDED_C5 backend %s rows %d touched %d weights %.2e (%s) points %.2e (%s)" % (backend, int(o1["counts"][3]), int(cap.item()), worst, wname, e, ename))
    assert worst < WTOL and e < PTOL, (worst, wname, e, ename)
    print("SHARDED_C5_OK")
dist.barrier()
dist.destroy_process_group()
'''


def _run_c5(tmp_path, backend, port):
    script = tmp_path / "w5.py"
    # 2-rank gradients == 1-rank gradients: identical per-row arithmetic, different summation order of the per-point / per-weight sums and another
    # power-of-two operand scale in the weight-gradient GEMMs (one per batch) -- fp32-rounding class, far below the tolerance against the reference
    script.write_text(_WORKER_C5.replace("WTOL", "2e-5").replace("PTOL", "2e-5"))       # measured: weights 1.2e-6, points 3.1e-7 of max
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", HNR_TEST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[0][-3000:] + outs[1][-3000:]
    assert "SHARDED_C5_OK" in outs[0], outs[0][-2000:]
    print([l for l in outs[0].splitlines() if l.startswith("SHARDED_C5 ")][0])


def test_c5_full_size_patch_sharded_train_step_with_blur_module(tmp_path):
    """BASELINE config C5 ("ScanNet livingroom train step with blur-handling module, 8 x MI355X (grad path + RCCL gather)") at the
    size SURVEY 8d states -- 49 dilated 8x8 patches (dilation_setup 7_8_1_6, data/scannet_ft_dataset.py:918-949), 12 symmetric 9x9
    blur kernels (:214-242), 2 M points, frame weight, both shipped loss terms -- through the PRODUCTION path bench.py times: two processes share
    the GPU, each runs train.train_step on its 24 / 25 whole patches, the gradients meet in parallel.allreduce_weight_grads (one all-reduce) and
    parallel.PointGradExchange (one all-gather; gloo on host copies here), and the summed gradients must equal the single-process train_step on
    all 49 patches.  Counterpart of models/base_rendering_model.py:677-745 + mvs_points_volumetric_model.py:111-148."""
    _run_c5(tmp_path, "gloo", 29643)


def test_c5_patch_sharded_train_step_over_rccl(tmp_path):
    """The same comparison with one GPU per rank and the two collectives on device tensors over RCCL (all_reduce + all_gather_into_tensor):
    runs where two GPUs are visible (round-5 advice: the no-host-read collective path needs hardware evidence), skipped on a 1-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the GPU box of this pool has one)")
    _run_c5(tmp_path, "nccl", 29645)
