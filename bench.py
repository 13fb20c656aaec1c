#!/usr/bin/env python3
"""Headline benchmark: rays/s of the forward hybrid render (query -> gather/aggregate -> composite)
on the scene0241_01-like synthetic config (BASELINE.json configs[2] / SURVEY.md section 8d C3).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full 620x460 = 285 200-ray frame of the FIXED scene0241_01-like ray batch (north_star): with N ranks the
frame's scan lines are dealt round-robin to the N ranks (parallel.shard_lines; --shard blocks: N contiguous blocks, whose work differs
by up to 1.68x on this frame, tools/shard_balance.py), every rank renders its rays (cloud,
grid, weights and reference-view features replicated and already resident in HBM) and the colours are reassembled on rank 0
with ONE RCCL gather -- strong scaling, value = 285 200 rays / max-over-ranks step time.  `--scaling weak` instead lets every
rank render a whole frame of its own (value = N x 285 200 / time).

`python bench.py --gpus N` with WORLD_SIZE unset spawns the N ranks itself (child processes, before anything touches the GPU);
under torch.distributed.run the launcher's RANK / LOCAL_RANK / WORLD_SIZE are used and must agree with --gpus.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_MFMA_PEAK_TF = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TF = 2500.0     # MI355X_MICROARCH.md: bf16 MFMA dense peak (v_mfma_f32_32x32x16_bf16)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--points", type=float, default=2.0e6)
    ap.add_argument("--chunk", type=int, default=0, help="rays per launch (0 = the whole frame in one launch)")
    ap.add_argument("--scene", default="scene0241")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--margin", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-leg", action="store_true")
    ap.add_argument("--train-sharded-only", action="store_true", help="of the training legs run only the C5 patch-sharded step (tools/predict_train_scaling.sh)")
    ap.add_argument("--no-f32-anchor", action="store_true", help="skip the one fp32-MFMA frame rendered beside the timed region (fp32_mfma_anchor)")
    ap.add_argument("--cpu-sample-rays", type=int, default=2304)
    ap.add_argument("--shard", choices=("lines", "blocks"), default="lines",
                    help="strong scaling: scan lines dealt round-robin to the ranks (balanced: busiest rank 1.01x the mean work at N = 8) or N "
                         "contiguous blocks of scan lines (the reference's chunk order; busiest block 1.68x the mean on this frame)")
    ap.add_argument("--knn-order", choices=("sorted", "reference"), default=None,
                    help="neighbour order of the query; default: whatever the library ships (HybridRenderer.knn_order = 'reference': slot for slot "
                         "the reference's insertion history, the order the training path uses too).  sorted = the reference's neighbour SETS in "
                         "ascending (d2, enumeration) order (hnr_query_params.knn_order = 1), an opt-in A/B")
    ap.add_argument("--band", type=int, default=1, help="--shard lines: scan lines per dealt band")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong: the N ranks share ONE fixed frame (north_star); weak: one whole frame per rank")
    ap.add_argument("--dump-colors", default="", help="rank 0 writes the assembled [R,3] colours of the last step to this .npy file")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N child ranks (fresh processes; this parent never touches the GPU
    and never exec()s), relay rank 0's output, fail if any rank fails."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    import tempfile
    procs = []
    with tempfile.TemporaryFile() as out0:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        # a rank that dies (an exception in a timed step) must not leave the others waiting in a collective until RCCL's watchdog gives up
        # (round-4 advice): the parent watches all of them and ends the survivors -- its own children, by PID -- as soon as one has failed
        while True:
            rcs = [p.poll() for p in procs]
            if all(rc is not None for rc in rcs):
                break
            if any(rc not in (None, 0) for rc in rcs):
                time.sleep(2.0)                                # (let the failing rank's neighbours fail by themselves first: their messages are the useful ones)
                for p in procs:
                    if p.poll() is None:
                        p.kill()
                rcs = [p.wait() for p in procs]
                break
            time.sleep(0.2)
        out0.seek(0)
        sys.stdout.write(out0.read().decode("utf-8", "replace"))
        sys.stdout.flush()
    if any(rcs):
        raise SystemExit("bench.py: rank exit codes %s" % rcs)


def build_world(args, dev, rank):
    from hybridneuralrendering_amd import scenes
    from hybridneuralrendering_amd.aggregator import PointAggregator
    from hybridneuralrendering_amd.render import HybridRenderer, PointCloud
    sc = scenes.make_scene(args.scene, int(args.points), 2, w=args.width, h=args.height)
    opt = sc.opt
    torch.manual_seed(0)
    agg = PointAggregator(opt)
    with torch.no_grad():          # random-init weights; scale the density head so opacities are spread over (0,1)
        agg.alpha_branch[0].weight.mul_(30.0)
        agg.alpha_branch[0].bias.fill_(30.0)
    agg = agg.to(dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cloud = PointCloud(t(sc.xyz), t(sc.emb), t(sc.conf), t(sc.dir), t(sc.color))
    rnd = HybridRenderer(opt, agg, dev)
    if getattr(args, "knn_order", None) and opt.K == 8:          # an explicit A/B only: the timed frame runs the library's default otherwise
        rnd.knn_order = args.knn_order
    # rank-specific camera: same scene, slightly different pose (weak scaling: every GPU renders a whole frame)
    eye = sc.c2w[:3, 3] + np.array([0.05, -0.04, 0.01], np.float32) * rank
    tgt = sc.c2w[:3, 3] + sc.c2w[:3, 2] * 3.0
    c2w = scenes.look_at(eye, tgt)
    pix = scenes.pixel_grid(sc.w, sc.h, args.margin)
    rays = scenes.camera_rays(pix, sc.intrinsic, c2w)
    cam = dict(raydir=t(rays), campos=t(c2w[:3, 3]), camrot=t(c2w[:3, :3]), bg=t(sc.bg_color),
               c2w_nearest=t(sc.c2w_nearest), campos_nearest=t(sc.c2w_nearest[:, :3, 3]), intrinsic=t(sc.intrinsic),
               images=t(sc.images_nearest), w2c_nearest=torch.inverse(t(sc.c2w_nearest)), c2w=c2w, pix=pix, rays_np=rays)
    return sc, opt, agg, cloud, rnd, cam


def render_frame(rnd, cloud, cam, sc, chunk, timers=None, statuses=None):
    R = cam["raydir"].shape[0]
    chunk = R if chunk <= 0 else chunk
    cols = []
    for lo in range(0, R, chunk):
        out = rnd.render_rays(cloud, cam["raydir"][lo:lo + chunk], cam["campos"], cam["camrot"], cam["bg"], sc.near, sc.far,
                              cam["c2w_nearest"], cam["campos_nearest"], cam["intrinsic"], cam["images"],
                              w2c_nearest=cam["w2c_nearest"], timers=timers)
        if statuses is not None and out.get("status") is not None:
            statuses.append(dict(status=out["status"]))
        cols.append(out["coarse_raycolor"])
    return cols[0] if len(cols) == 1 else torch.cat(cols, dim=0), out


def _newest_profile(suffix):
    """profiles/rNN_<suffix> of the latest round that has one (the PMC passes are re-collected when the kernels change: tools/gpu_job.sh)"""
    for tag in ("r05", "r04", "r03"):
        if os.path.exists(os.path.join(ROOT, "profiles", "%s_%s" % (tag, suffix))):
            return "%s_%s" % (tag, suffix)
    return "r05_" + suffix


TRAFFIC_JSON = _newest_profile("traffic.json")
TRAIN_TRAFFIC_JSON = _newest_profile("train_traffic.json")
CHAIN_PMC_JSON = _newest_profile("chain_pmc.json")


def pmc_traffic(name=TRAFFIC_JSON):
    """HBM bytes per launch from the rocprofv3 PMC passes kept under profiles/ (FETCH_SIZE / WRITE_SIZE cannot be read
    from inside the process; the passes are re-collected with tools/collect_traffic.py / collect_train_traffic.py whenever the kernels change)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))["kernels"]
    except Exception:
        return {}
    return d


def _oracle_pass(sc, opt, sd, rays, c2w):
    """One pass of the CPU oracle over a ray batch: C query restatement (grid build included -- the reference rebuilds its grid for
    every chunk) + torch-CPU gather / aggregate / composite.  Returns (colours [n,3], seconds, query seconds)."""
    from oracle import query_oracle as qo, render_oracle as ro
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    t0 = time.time()
    hp = qo.hyperparameters(sc.xyz, opt.vsize, opt.vscale, opt.kernel_size, opt.ranges, opt.radius_limit_scale)
    g = qo.OracleGrid(sc.xyz, hp["origin"], hp["cell"], hp["dims"], opt.query_size, opt.P, opt.max_o)
    q = g.query(c2w[:3, 3], rays, qo.tmid_table(sc.near, sc.far, opt.z_depth_dim), opt.SR, opt.K, hp["radius2"], opt.kernel_size)
    t_query = time.time() - t0
    with torch.no_grad():
        ref = ro.render(t(sc.xyz), t(sc.emb), t(sc.conf), t(sc.dir), t(sc.color), sd, q, t(c2w[:3, 3])[None], t(c2w[:3, :3])[None],
                        t(rays)[None], t(sc.bg_color)[None], t(sc.c2w_nearest)[None], t(sc.c2w_nearest[:, :3, 3])[None],
                        t(sc.intrinsic)[None], t(sc.images_nearest)[None], opt.vsize)
    return ref["full_coarse_raycolor"][0].numpy(), time.time() - t0, t_query


def cpu_baseline(args, sc, opt, agg, cam, gpu_colors):
    """the CPU oracle (a port: C query restatement + torch-CPU aggregate / composite, pinned to the imported reference by the
    golden fixtures) timed on this box's host cores, 1 warm-up + 3 timed passes each, on
      * C3: one 48x48 = 2304-ray chunk of the SAME frame the GPU renders (the reference's evaluation chunk, run/test_ft.py:325), and
      * C1: the chair 200x200 camera, one 32x32 = 1024-ray batch (100 k points, SR 80, P 12; dev_scripts/w_n360/chair_hybrid.sh).
    `value` is the C3 rate (same workload as the headline metric); the C1 rate is reported beside it."""
    from hybridneuralrendering_amd import scenes
    from hybridneuralrendering_amd.aggregator import PointAggregator
    n = args.cpu_sample_rays
    side = int(np.sqrt(n))
    W = sc.w - 2 * args.margin
    H = sc.h - 2 * args.margin
    x0, y0 = (W - side) // 2, (H - side) // 2
    idx = ((y0 + np.arange(side))[:, None] * W + (x0 + np.arange(side))[None, :]).reshape(-1)
    rays = cam["rays_np"][idx]
    sd = {k: v.detach().cpu() for k, v in agg.state_dict().items()}
    cores = torch.get_num_threads()
    times, tq = [], 0.0
    for it in range(4):                                   # 1 warm-up + 3 timed
        refc, dt, tq = _oracle_pass(sc, opt, sd, rays, cam["c2w"])
        if it > 0:
            times.append(dt)
    got = gpu_colors[idx]
    mse = float(np.mean((refc.astype(np.float64) - got.astype(np.float64)) ** 2))
    psnr = 99.0 if mse == 0 else -10.0 * np.log10(mse)
    dt3 = float(np.mean(times))
    # the error bound over the WHOLE frame, not one block: further 48x48 blocks spread over the frame (corners, edges, between), same oracle (one grid
    # build for all of them: these passes are checks, not timings)
    from oracle import query_oracle as qo, render_oracle as ro
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    hp = qo.hyperparameters(sc.xyz, opt.vsize, opt.vscale, opt.kernel_size, opt.ranges, opt.radius_limit_scale)
    og = qo.OracleGrid(sc.xyz, hp["origin"], hp["cell"], hp["dims"], opt.query_size, opt.P, opt.max_o)
    tm = qo.tmid_table(sc.near, sc.far, opt.z_depth_dim)
    # Beside the fp32 oracle, the SAME oracle (same neighbour sets) evaluated in fp64: the reference truncates the reprojected pixel coordinates
    # (point_aggregators.py:1077-1078), so a one-ulp difference in the 4x4 inverse or the projection (torch's BLAS / LAPACK on the CPU, explicit
    # fp32 multiply-adds on the GPU) moves a gathered feature to the neighbouring pixel on a few rays -- a discrete change of ~1e-4 that any two
    # fp32 evaluations of the reference can show.  fp32-vs-fp64 of the oracle itself is the yardstick for it.
    def block_render(bi, q, dt):
        t2 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dt) if np.asarray(a).dtype.kind == "f" else torch.from_numpy(np.ascontiguousarray(a))
        sdd = {k: (v.to(dt) if v.dtype.is_floating_point else v) for k, v in sd.items()}
        if isinstance(q, dict):
            q = {k: (torch.as_tensor(v).to(dt) if isinstance(v, (np.ndarray, torch.Tensor)) and torch.as_tensor(v).dtype.is_floating_point else v) for k, v in q.items()}
        torch.set_default_dtype(dt)
        try:
            with torch.no_grad():
                return ro.render(t2(sc.xyz), t2(sc.emb), t2(sc.conf), t2(sc.dir), t2(sc.color), sdd, q, t2(cam["c2w"][:3, 3])[None], t2(cam["c2w"][:3, :3])[None],
                                 t2(cam["rays_np"][bi])[None], t2(sc.bg_color)[None], t2(sc.c2w_nearest)[None], t2(sc.c2w_nearest[:, :3, 3])[None],
                                 t2(sc.intrinsic)[None], t2(sc.images_nearest)[None], opt.vsize)["full_coarse_raycolor"][0].numpy().astype(np.float64)
        finally:
            torch.set_default_dtype(torch.float32)
    from hybridneuralrendering_amd import _lib
    Lh, dev = _lib.lib(), cam["w2c_nearest"].device
    Hn, Wn = int(cam["images"].shape[-3]), int(cam["images"].shape[-2])
    def pixel_agreement(q):
        """per VALID ray of the block: every valid sample's pixel in every view is the same in the oracle and on the GPU"""
        loc = np.ascontiguousarray(q["sample_loc_w"], np.float32)                       # [R', SR, 3]
        valid = (np.asarray(q["sample_pidx"]) >= 0).any(axis=-1)                        # [R', SR]
        po = ro.gathered_pixels(tt(loc), tt(sc.c2w_nearest)[None], tt(sc.intrinsic)[None], Hn, Wn).numpy()           # [V, R', SR, 2]
        n = loc.shape[0] * loc.shape[1]
        d_loc = torch.from_numpy(loc.reshape(-1, 3)).to(dev); d_item = torch.arange(n, dtype=torch.int32, device=dev)
        d_cnt = torch.zeros((16,), dtype=torch.int64, device=dev); d_cnt[_lib.CNT["SAMPLES_VALID"]] = n
        V = int(cam["w2c_nearest"].shape[0])
        d_pix = torch.full((V, n, 2), -7, dtype=torch.int32, device=dev)
        _lib.check(Lh.hnr_proj_pixels(_lib.ptr(d_loc), _lib.ptr(d_item), _lib.ptr(d_cnt), _lib.ptr(cam["w2c_nearest"].contiguous()), _lib.ptr(cam["intrinsic"].contiguous()),
                                      V, Hn, Wn, n, _lib.ptr(d_pix), _lib.stream()), "hnr_proj_pixels")
        pg = d_pix.cpu().numpy().reshape(V, loc.shape[0], loc.shape[1], 2)
        diff = ((pg != po).any(axis=-1) & valid[None]).any(axis=0)                      # [R', SR]
        return ~diff.any(axis=1)
    blocks, all_err, all_same = [], [], []
    for fx, fy in ((0.5, 0.5), (0.0, 0.0), (1.0, 0.0), (0.0, 1.0), (1.0, 1.0), (0.5, 0.05), (0.25, 0.6), (0.8, 0.35)):
        bx, by = (int(x0), int(y0)) if (fx, fy) == (0.5, 0.5) else (int(fx * (W - side)), int(fy * (H - side)))
        bi = ((by + np.arange(side))[:, None] * W + (bx + np.arange(side))[None, :]).reshape(-1)
        q = og.query(cam["c2w"][:3, 3], cam["rays_np"][bi], tm, opt.SR, opt.K, hp["radius2"], opt.kernel_size)
        rb, rb64 = block_render(bi, q, torch.float32), block_render(bi, q, torch.float64)
        gb = gpu_colors[bi].astype(np.float64)
        err = np.abs(rb - gb).max(axis=1)
        all_err.append(err)
        # which rays gather the SAME reference-view pixels in both evaluations: the oracle's truncated projections (its own torch ops) against the
        # pixels the HIP merge stage gathers (hnr_proj_pixels: the device function the merge kernels call, on the very positions -- the query is bit-exact)
        pix_same = pixel_agreement(q)
        same_mask = np.zeros(len(bi), bool); same_mask[np.flatnonzero(np.asarray(q["ray_mask"]) > 0)] = pix_same; same_mask[np.asarray(q["ray_mask"]) == 0] = True
        all_same.append(same_mask)
        m2 = float(np.mean((rb - gb) ** 2))
        blocks.append(dict(x0=bx, y0=by, max_abs=float(err.max()), psnr_db=round(99.0 if m2 == 0 else -10.0 * np.log10(m2), 2),
                           rays_over_1e_4=int((err > 1e-4).sum()), rays_with_another_pixel=int((~same_mask).sum()),
                           max_abs_same_pixels=float(err[same_mask].max()) if same_mask.any() else 0.0,
                           max_abs_other_pixel=float(err[~same_mask].max()) if (~same_mask).any() else 0.0, oracle_f32_vs_f64_max_abs=float(np.abs(rb - rb64).max()),
                           oracle_rays_over_1e_4=int((np.abs(rb - rb64).max(axis=1) > 1e-4).sum())))
    all_err = np.concatenate(all_err); all_same = np.concatenate(all_same)
    worst = max(b["max_abs"] for b in blocks)
    worst_same = float(all_err[all_same].max()) if all_same.any() else 0.0
    # the stated tolerance is ASSERTED on every ray whose gathered pixels agree; a ray that gathers another pixel than the oracle in some view is a discrete
    # difference of the reference's truncation rule, reported (count + its largest error), not an arithmetic error
    if not worst_same <= 1e-4:
        raise SystemExit("bench.py: GPU frame differs from the CPU oracle by %.3e (> 1e-4) on a ray whose reference-view pixels agree" % worst_same)
    # ... and the rays left out of that assertion are bounded too (round-4 advice: a systematic projection error would move most rays into this set): a
    # ray gathers another pixel only when a sample sits within an ulp of a pixel border -- the oracle's own fp32 and fp64 evaluations disagree on a
    # handful of rays of 18 432 for the same reason -- and its colour then moves by one pixel's worth of one view's feature, not arbitrarily
    n_other = int((~all_same).sum())
    worst_other = float(all_err[~all_same].max()) if n_other else 0.0
    if n_other > max(64, int(0.005 * all_err.size)) or worst_other > 5e-3:
        raise SystemExit("bench.py: %d of %d checked rays gather another reference-view pixel than the oracle (max |d| %.3e): more than pixel-border ties explain"
                         % (n_other, all_err.size, worst_other))
    # C1
    sc1 = scenes.make_scene("chair", 100000, 0)
    sc1.opt.agg_axis_weight = None
    px, py = np.meshgrid(np.arange(84, 116), np.arange(84, 116), indexing="ij")
    rays1 = scenes.camera_rays(np.stack([px, py], axis=-1).reshape(-1, 2).astype(np.int32), sc1.intrinsic, sc1.c2w)
    t1 = []
    for it in range(4):
        _, dt, _ = _oracle_pass(sc1, sc1.opt, sd, rays1, sc1.c2w)
        if it > 0:
            t1.append(dt)
    return dict(value=len(idx) / dt3, unit="rays/s", cores=cores, kind="port",
                sample="C3: one %dx%d-ray chunk of the same frame, 1 warm-up + 3 timed passes (%.2f s each): C oracle grid build over %d points + "
                       "query (%.2f s, 1 thread) + torch-CPU aggregate/composite with 4 reference views (%d threads)" % (
                           side, side, dt3, sc.xyz.shape[0], tq, cores),
                c1_chair=dict(value=round(rays1.shape[0] / float(np.mean(t1)), 1), unit="rays/s",
                              sample="C1: chair 200x200 camera, one 32x32 = 1024-ray batch, 100 k points, SR 80, P 12; 1 warm-up + 3 timed passes "
                                     "(%.2f s each)" % float(np.mean(t1))),
                psnr_gpu_vs_oracle_db=round(min(b["psnr_db"] for b in blocks), 2), max_abs_gpu_vs_oracle=worst,
                max_abs_gpu_vs_oracle_same_pixels=worst_same, rays_gathering_another_pixel=int((~all_same).sum()),
                max_abs_on_rays_gathering_another_pixel=float(all_err[~all_same].max()) if (~all_same).any() else 0.0,
                asserted="max-abs <= 1e-4 on every checked ray whose gathered reference-view pixels equal the oracle's (hnr_proj_pixels vs oracle.gathered_pixels); "
                         "the other rays: at most max(64, 0.5 %) of the checked ones, max-abs <= 5e-3",
                rays_checked=int(all_err.size), rays_over_1e_4=int((all_err > 1e-4).sum()), p999_abs_gpu_vs_oracle=float(np.quantile(all_err, 0.999)),
                oracle_f32_vs_f64_max_abs=max(b["oracle_f32_vs_f64_max_abs"] for b in blocks), oracle_rays_over_1e_4=sum(b["oracle_rays_over_1e_4"] for b in blocks),
                checked_blocks=blocks, tolerance="fp32 max-abs <= 1e-4 on coarse_raycolor (SURVEY 8d) over %d blocks of %dx%d rays spread over the frame, except on rays "
                "where a reprojected sample sits within an ulp of a pixel boundary (the reference truncates the coordinate: the gathered pixel is then decided by the "
                "rounding of the 4x4 inverse / projection; oracle_f32_vs_f64_* = the same effect between two evaluations of the oracle itself)" % (len(blocks), side, side))


def train_leg_sharded(args, sc, opt, agg, cloud, rnd, cam, dev, world, rank, rehearsal, emulate, steps=20, warmup=3):
    """BASELINE config C5 the way it runs on N GPUs (models/mvs_points_volumetric_model.py:111-152, models/base_rendering_model.py:677-745): one batch of
    49 dilated 8x8 patches (3136 rays, dilation_setup 7_8_1_6), the blur-handling module (12 symmetric 9x9 kernels) and the item's frame weight; the batch is
    sharded by WHOLE patches (parallel.shard_patches), every rank runs forward -> blur module -> loss kernels -> blur backward -> backward on its 6-7 patches
    (train.train_step: no autograd graph, no torch.unique, no host read; HNR_BENCH_TRAIN_GRAPH=1: captured in a hipGraph and replayed), then the gradients
    meet in TWO collectives without a host read: ONE all-reduce of the flat weight-gradient buffer carrying the ranks' valid-ray counts
    (parallel.allreduce_weight_grads: the loss is a mean over the batch's valid rays) and ONE fixed-capacity all-gather of packed (point id | 39 floats)
    records of the touched points (parallel.PointGradExchange; the touched list is what the forward call left on the device).
    Timed with HIP events per part; max over ranks.  world == 1 and HNR_BENCH_EMULATE_RANK=r/n: rank r's share of an n-way split alone on this GPU -- the
    collectives degenerate to their local pack / apply parts, which are still run and timed (tools/predict_train_scaling.sh)."""
    import torch.distributed as dist
    from hybridneuralrendering_amd import scenes, parallel
    from hybridneuralrendering_amd.train import TrainPath, train_step, CapturedTrainStep
    old_train, old_dil = opt.is_train, getattr(opt, "dilation_setup", None)
    opt.is_train, opt.dilation_setup = 1, "7_8_1_6"
    # HNR_BENCH_TRAIN_GRAPH=1: replay the step from a hipGraph (train.CapturedTrainStep).  Measured in round 5 and NOT the default: the ROCm 7.2 graph
    # executor runs the step's three queues one after the other (a 1/8 share: 2.60 ms replayed = the single-queue eager step, 2.18 ms eager with the side streams)
    use_graph = os.environ.get("HNR_BENCH_TRAIN_GRAPH", "0") == "1"
    try:
        pix, pn, ps = scenes.dilated_patch_batch(sc.w, sc.h, args.margin, opt.dilation_setup, seed=4)
        S = pn * ps
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        rays_all = t(scenes.camera_rays(pix, sc.intrinsic, sc.c2w))
        kern = t(scenes.blur_kernels_v2())[None]
        g = torch.Generator().manual_seed(9)
        gt = torch.rand((S * S, 3), generator=g).to(dev)
        drop = parallel.global_drop_flags(pn, ps, opt.drop_ratio).to(dev)
        frame_weight = 0.7
        n_way, r_of = (world, rank)
        if emulate and world == 1:
            r_of, n_way = (int(x) for x in emulate.split("/"))
        ids, ray_ids = parallel.shard_patches(pn, ps, n_way, r_of)
        ray_ids = (torch.arange(S * S) if n_way == 1 else ray_ids).to(dev)       # one rank: the batch in its own (row-major) order
        layout, n_patches = ("grid", pn) if n_way == 1 else ("patch_major", int(ids.numel()))
        path = TrainPath(rnd)
        path.reuse_outputs = True                                # a training loop: every step writes the same output / gradient tensors
        leaves = [x.clone().requires_grad_(True) for x in (cloud.emb, cloud.conf, cloud.dir, cloud.color)]
        for prm in agg.parameters():
            prm.requires_grad_(True)
        my_rays, my_gt, my_drop = rays_all[ray_ids].contiguous(), gt[ray_ids].contiguous(), drop[ray_ids].contiguous()
        w2c = torch.inverse(cam["c2w_nearest"]).contiguous()
        ev = lambda: torch.cuda.Event(enable_timing=True)
        state = dict(cap=None, ex=None)

        def compute():
            """one rank's step: eager (train_step) or a replay of the captured graph; the jittered depth tables are drawn inside either way"""
            if state["cap"] is not None:
                return state["cap"].step(assign_grads=False)
            return train_step(path, agg, cloud.xyz, leaves[0], leaves[1], leaves[2], leaves[3], my_rays, cam["campos"], cam["camrot"], cam["bg"], sc.near, sc.far,
                              cam["c2w_nearest"], cam["campos_nearest"], cam["intrinsic"], cam["images"], my_gt, zero_epsilon=1e-3, w_color=1.0, w_zero_one=1e-4,
                              frame_weight=frame_weight, ray_drop=my_drop, assign_grads=False, blur_kernels=kern, patch_num=n_patches, patch_size=ps,
                              patch_layout=layout, w2c_nearest=w2c)

        def one(timed=None, collect=True):
            e = [ev() for _ in range(4)] if timed is not None else None
            if e: e[0].record()
            out, pg, ag = compute()
            if e: e[1].record()
            Sv = out["_saved"]
            nv = out["loss"][3:4]
            bufs = [pg["points_embeding"], pg["points_conf"], pg["points_dir"], pg["points_color"]]
            if collect and state["ex"] is not None:
                if rehearsal and world > 1:                                     # gloo on host copies: control flow only
                    flat = Sv.flat.cpu()
                    parallel.allreduce_weight_grads(flat, nv.cpu(), Sv.flat_payload)
                    tids, tcnt = TrainPath.touched_points(Sv)
                    hb = [b.cpu() for b in bufs]
                    rec = state["ex"].pack(hb, tids.cpu(), tcnt.cpu(), nv.cpu())
                    state["ex"].apply(state["ex"].exchange(rec), hb, rank)
                    if e: e[2].record()
                else:
                    parallel.allreduce_weight_grads(Sv.flat, nv, Sv.flat_payload)
                    if e: e[2].record()
                    tids, tcnt = TrainPath.touched_points(Sv)
                    rec = state["ex"].pack(bufs, tids, tcnt, nv)
                    _tot, over = state["ex"].apply(state["ex"].exchange(rec), bufs, rank if world > 1 else 0)
                    state["over"] = over if "over" not in state else torch.maximum(state["over"], over)    # any step of the loop (the jitter changes the touched set)
            elif e:
                e[2].record()
            if e: e[3].record()
            if timed is not None: timed.append(e)
            return out
        # Preflight: this leg is reported BESIDE the headline line, so it must not be able to take the run down or leave ranks waiting in a collective
        # for one that raised.  Every rank runs one eager step (and, by default, captures the step in a hipGraph) without the collectives, the ranks agree
        # on the outcome and on the exchange capacity (one all-reduce that every rank reaches), and only then the collectives run.
        err, n_touched, graph_note = None, 0, "eager: train.train_step, launches queued back to back on three queues"
        try:
            out = one(collect=False)
            torch.cuda.synchronize()
            TrainPath.check_status(out)
            n_touched = int(TrainPath.touched_points(out["_saved"])[1].item())
            if use_graph:
                try:
                    sample = dict(raydir=my_rays, campos=cam["campos"], camrot=cam["camrot"], bg_color=cam["bg"], c2w_nearest=cam["c2w_nearest"], w2c_nearest=w2c,
                                  campos_nearest=cam["campos_nearest"], intrinsic_nearest=cam["intrinsic"], images_nearest=cam["images"], gt_image=my_gt,
                                  ray_drop=my_drop, blur_kernels=kern, frame_weight=frame_weight)
                    state["cap"] = CapturedTrainStep(path, agg, cloud.xyz, leaves[0], leaves[1], leaves[2], leaves[3], sample, sc.near, sc.far, zero_epsilon=1e-3,
                                                     w_color=1.0, w_zero_one=1e-4, patch_num=n_patches, patch_size=ps, patch_layout=layout)
                    graph_note = "hipGraph replay (train.CapturedTrainStep)"
                except Exception as ex:                                          # noqa: BLE001  (the eager step is the fallback of the MEASUREMENT, not of the product)
                    if os.environ.get("HNR_BENCH_STRICT"):
                        raise
                    state["cap"] = None
                    graph_note = "eager: capture failed (%s: %s)" % (type(ex).__name__, str(ex)[:200])
        except Exception as ex:                                                  # noqa: BLE001
            if os.environ.get("HNR_BENCH_STRICT"):
                raise
            err = "%s: %s" % (type(ex).__name__, str(ex)[:300])
        flag = torch.tensor([0 if err else 1, -n_touched], dtype=torch.int64, device="cpu" if (rehearsal or world == 1) else dev)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag[0].item()) == 0 and err is None:
                err = "another rank failed its preflight step"
        if err:
            return dict(workload="C5 sharded train step", error=err, n_ranks=n_way)
        capacity = max(1024, (int(-flag[1].item()) * 2 + 255) // 256 * 256)      # 2 x the busiest rank's touched points of the preflight step
        state["ex"] = parallel.PointGradExchange(capacity) if n_way > 1 else None     # (one rank, nothing emulated: there is nothing to exchange)
        try:
            for _ in range(warmup):
                out = one()
            if world > 1: dist.barrier()
            torch.cuda.synchronize()
            evs = []
            t0 = time.perf_counter()
            for _ in range(steps):
                out = one(evs)
            if world > 1: dist.barrier()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
        except Exception as ex:                                                  # noqa: BLE001
            if os.environ.get("HNR_BENCH_STRICT") or world > 1:                  # (N > 1: the other ranks wait in a collective -- fail the run rather than hang it)
                raise
            return dict(workload="C5 sharded train step", error="%s: %s" % (type(ex).__name__, str(ex)[:300]), n_ranks=n_way)
        comp = sum(e[0].elapsed_time(e[1]) for e in evs) / steps
        ar_w = sum(e[1].elapsed_time(e[2]) for e in evs) / steps
        ar_p = sum(e[2].elapsed_time(e[3]) for e in evs) / steps
        per_rank = [dt * 1e3]
        tt = torch.tensor([dt, comp * 1e-3, ar_w * 1e-3, ar_p * 1e-3], dtype=torch.float64, device=dev)
        if world > 1:
            tt = tt.cpu() if rehearsal else tt
            allt = [torch.empty_like(tt) for _ in range(world)]
            dist.all_gather(allt, tt)
            per_rank = [round(float(x[0]) * 1e3, 3) for x in allt]
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, comp, ar_w, ar_p = (float(x) for x in tt)
        c = out["counts"].cpu().numpy()
        over = float(state.get("over", torch.zeros(())).item()) if "over" in state else 0.0
        n_w = int(out["_saved"].flat_payload)
        if over:
            # a rank touched more points than the agreed capacity in some step: its extra rows stayed local, the replicas would diverge -- not a valid timing
            return dict(workload="C5 sharded train step", n_ranks=n_way, exchange_capacity=capacity,
                        error="PointGradExchange overflowed its capacity of %d records in at least one timed step" % capacity)
        return dict(workload="C5: %d dilated %dx%d patches (dilation_setup 7_8_1_6) = %d rays, blur module (12 kernels 9x9) + frame weight, fwd + bwd%s" % (
                        pn * pn, ps, ps, S * S, "" if n_way == 1 else "; rank %d of %d: %d patches = %d rays" % (r_of, n_way, int(ids.numel()), int(ray_ids.numel()))),
                    ms_per_step=round(dt * 1e3, 3), compute_ms=round(comp * 1e3, 3), allreduce_weights_ms=round(ar_w * 1e3, 3), exchange_points_ms=round(ar_p * 1e3, 3),
                    per_rank_ms_per_step=per_rank, n_ranks=n_way, rccl_ranks=(world if (world > 1 and not rehearsal) else 0), steps=steps, step_form=graph_note,
                    emulated_rank=("%d/%d on one GPU: the collectives are their local pack / apply parts only" % (r_of, n_way)) if (emulate and world == 1) else None,
                    valid_samples=int(c[6]), neighbour_rows=int(c[3]), touched_points=n_touched, exchange_capacity=capacity, exchange_overflow=bool(over),
                    collective_bytes=dict(weights_allreduce=4 * (n_w + 1), point_records_allgather_per_rank=(capacity + 2) * 40 * 4,
                                          dense_point_allreduce_avoided=int(sum(x.numel() for x in leaves) * 4)),
                    note="max over ranks; no host read in the step; collectives: parallel.allreduce_weight_grads (ONE all-reduce of the flat weight-gradient buffer + the "
                         "valid-ray count) and parallel.PointGradExchange (ONE fixed-capacity all-gather of packed (id | 39 floats) records, applied in rank order)")
    finally:
        opt.is_train, opt.dilation_setup = old_train, old_dil
        for prm in agg.parameters():
            prm.requires_grad_(False)


def train_leg(args, sc, opt, agg, cloud, rnd, cam, dev, steps=20, warmup=3):
    """SURVEY 8d config C3 (fwd+bwd): one 56x56 = 3136-ray training batch (random window, jittered depths, patch drop) through
    the HIP forward + backward with the shipped loss terms.  Reported beside the headline metric, never part of `value`."""
    from hybridneuralrendering_amd import scenes
    from hybridneuralrendering_amd.train import TrainPath, train_step
    old = opt.is_train
    opt.is_train = 1
    try:
        path = TrainPath(rnd)
        path.reuse_outputs = True                                # a training loop: every step writes the same output / gradient tensors
        rng = np.random.default_rng(17)
        x0 = int(rng.integers(args.margin, sc.w - args.margin - 56)); y0 = int(rng.integers(args.margin, sc.h - args.margin - 56))
        px, py = np.meshgrid(np.arange(x0, x0 + 56), np.arange(y0, y0 + 56), indexing="ij")
        pix = np.stack([px, py], axis=-1).reshape(-1, 2).astype(np.int32)
        raydir = torch.from_numpy(scenes.camera_rays(pix, sc.intrinsic, sc.c2w)).to(dev)
        gt = torch.rand((raydir.shape[0], 3), device=dev)
        w2c_c3 = torch.inverse(cam["c2w_nearest"]).contiguous()   # the item's reference-view poses inverted once per item (four 4x4 matrices: data-loader work)
        leaves = [t.clone().requires_grad_(True) for t in (cloud.emb, cloud.conf, cloud.dir, cloud.color)]
        for prm in agg.parameters():
            prm.requires_grad_(True)
        def one(ev=None):
            for t in leaves:
                t.grad = None
            agg.zero_grad(set_to_none=True)
            if ev: ev[0].record()
            # forward -> the shipped loss terms (masked colour MSE + zero-one regulariser on conf_coefficient of the valid rays; value and
            # gradients on the device, hnr_shipped_loss_rows) -> backward, queued back to back (train.train_step): no autograd graph, no host read
            out, _pg, _ag = train_step(path, agg, cloud.xyz, leaves[0], leaves[1], leaves[2], leaves[3], raydir, cam["campos"], cam["camrot"],
                                       cam["bg"], sc.near, sc.far, cam["c2w_nearest"], cam["campos_nearest"], cam["intrinsic"], cam["images"], gt,
                                       zero_epsilon=1e-3, w_color=1.0, w_zero_one=1e-4, w2c_nearest=w2c_c3)
            if ev: ev[1].record()
            return out
        for _ in range(warmup):
            out = one()
        torch.cuda.synchronize()
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(steps)]
        t0 = time.perf_counter()
        for i in range(steps):
            out = one(evs[i])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        gpu_ms = sum(e[0].elapsed_time(e[1]) for e in evs) / steps
        c = out["counts"].cpu().numpy()
        # the same step captured once in a hipGraph and replayed (train.CapturedTrainStep; the depth jitter is drawn inside the graph), reported beside the
        # eager number: on ROCm 7.2 the graph executor serialises the step's three queues, so the replay is the SLOWER form
        captured_ms, graph_note = None, "not measured (HNR_BENCH_TRAIN_GRAPH=0)"
        if os.environ.get("HNR_BENCH_TRAIN_GRAPH", "1") != "0":
            try:
                from hybridneuralrendering_amd.train import CapturedTrainStep
                sample = dict(raydir=raydir, campos=cam["campos"], camrot=cam["camrot"], bg_color=cam["bg"], c2w_nearest=cam["c2w_nearest"],
                              campos_nearest=cam["campos_nearest"], intrinsic_nearest=cam["intrinsic"], images_nearest=cam["images"], gt_image=gt)
                capt = CapturedTrainStep(path, agg, cloud.xyz, leaves[0], leaves[1], leaves[2], leaves[3], sample, sc.near, sc.far, zero_epsilon=1e-3,
                                         w_color=1.0, w_zero_one=1e-4)
                for _ in range(warmup):
                    capt.step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(steps):
                    out_c, _, _ = capt.step()
                torch.cuda.synchronize()
                captured_ms = round((time.perf_counter() - t0) / steps * 1e3, 3)
                TrainPath.check_status(out_c)
                graph_note = "hipGraph replay of the same launches (train.CapturedTrainStep)"
                del capt
            except Exception as ex:                                              # noqa: BLE001
                if os.environ.get("HNR_BENCH_STRICT"):
                    raise
                graph_note = "capture failed (%s: %s)" % (type(ex).__name__, str(ex)[:200])
        # stage times of the two library calls (HIP events recorded by the library at its stage boundaries, one extra step)
        path.timers = {}
        one()
        torch.cuda.synchronize()
        stage = {("fwd." + k): round(v, 4) for k, v in path.timers["fwd"][0].elapsed_ms().items()}
        stage.update({("bwd." + k): round(v, 4) for k, v in path.timers["bwd"][0].elapsed_ms().items()})
        path.timers = None
        fwd = sum(v for k, v in stage.items() if k.startswith("fwd."))          # the forward call's share (its stage events); the rest: loss kernels + backward
        bwd = gpu_ms - fwd
        # roofline of the step's dominant kernel: the weight-gradient GEMM dW = dZ^T X of a 256 x 256 per-neighbour layer (hnr_h2wgrad, five such
        # launches per step), timed alone with HIP events on tensors of the step's row count (8 row slots per valid sample)
        from hybridneuralrendering_amd import _lib
        Lh = _lib.lib()
        M8 = 8 * int(c[6])
        Zt, Xt = torch.randn((max(M8, 1), 256), device=dev), torch.randn((max(M8, 1), 256), device=dev)
        mz = torch.tensor([np.float32(8.0).view(np.int32)], dtype=torch.int32, device=dev)
        scr = torch.empty((int(Lh.hnr_h2wgrad_scratch_bytes(256, 256)),), dtype=torch.uint8, device=dev)
        dW, db = torch.empty((256, 256), device=dev), torch.empty((256,), device=dev)
        def wg():
            _lib.check(Lh.hnr_h2wgrad(_lib.ptr(Zt), 256, _lib.ptr(Xt), 256, M8, None, 1, 0, 256, 256, _lib.ptr(mz), _lib.ptr(mz), _lib.ptr(dW), 256, _lib.ptr(db), 0,
                                      _lib.ptr(scr), _lib.stream()), "hnr_h2wgrad")
        for _ in range(3):
            wg()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            wg()
        e1.record(); torch.cuda.synchronize()
        ms_w = e0.elapsed_time(e1) / 10
        issued = 3.0 * 2.0 * M8 * 256 * 288                                  # 3 fp16 MFMAs per fp32 product, K + 1 (bias column) padded to 9 tiles of 32
        # HBM bytes of the same kernel inside the step (PMC passes over tools/probe_train.py; only valid for the default C3 batch: 307 120 row slots)
        # PMC bytes of the 256-wide weight gradient (its launches inside the training step, profiles/<TRAIN_TRAFFIC_JSON>), selected by kernel name;
        # the batch behind that file is this one up to the depth jitter (row slots within 1 %: `traffic_rows` beside it)
        t_wg = [v for k, v in pmc_traffic(TRAIN_TRAFFIC_JSON).items() if "h2wgrad_dma_kernel" in k or "h2wgrad_kernel<8, 9" in k]   # (the DMA-staged kernel is the default since round 4)
        roof_t = dict(kernel="h2wgrad_dma_kernel + reduce (hnr_h2wgrad: dW = dZ^T X, db of one 256 x 256 per-neighbour layer; M = %d row slots)" % M8, bound="hbm",
                      achieved=round(M8 * 2048.0 / (ms_w * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(M8 * 2048.0 / (ms_w * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      traffic=int(t_wg[0]["hbm_bytes"]) if t_wg else None,
                      traffic_source=("profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE over tools/probe_train.py, bytes per launch: operands + %d KiB of "
                                      "per-workgroup partial sums)" % (TRAIN_TRAFFIC_JSON, 256 * 288 * 4 * 256 // 1024)) if t_wg else None,
                      avg_launch_ms=round(ms_w, 4), algorithmic_bytes_per_launch=int(M8 * 2048),
                      mfma_tflops_issued=round(issued / (ms_w * 1e-3) / 1e12, 1), fp32_equivalent_tflops=round(2.0 * M8 * 256 * 256 / (ms_w * 1e-3) / 1e12, 1),
                      note="algorithmic bytes = the two fp32 operands read once (2 KiB per row); the f16x2 MFMA work of this shape (3 x 2 M N K) would take "
                           "%.3f ms at the 2.5 PFLOP/s peak, the operand stream %.3f ms at 8 TB/s: HBM is the nearer roof" % (issued / 2.5e15 * 1e3, M8 * 2048.0 / 8e12 * 1e3)) if M8 > 0 else None
        return dict(workload="C3: 56x56 = %d rays, fwd (train mode) + bwd, shipped loss" % raydir.shape[0], ms_per_step=round(dt * 1e3, 3),
                    captured_ms_per_step=captured_ms, captured_form=graph_note, rays_per_s=round(raydir.shape[0] / dt, 1), fwd_ms=round(fwd, 3), loss_bwd_ms=round(bwd, 3),
                    neighbour_rows=int(c[3]), valid_samples=int(c[6]), steps=steps, entry="hnr_render_train_forward + hnr_render_train_backward (two library calls per step, no host read)",
                    stage_ms=stage, roofline_train=roof_t)
    finally:
        opt.is_train = old
        for prm in agg.parameters():
            prm.requires_grad_(False)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args)              # the parent has not touched the GPU (device_count() / is_available() not called yet)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher set WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # Rehearsal of the multi-rank control flow on a box with fewer GPUs than ranks (HNR_BENCH_REHEARSAL=1 only: ranks share
    # devices and the collectives run over gloo on host copies -- numbers from such a run mean nothing and say so).
    rehearsal = world > 1 and os.environ.get("HNR_BENCH_REHEARSAL") == "1" and torch.cuda.device_count() < world
    if world > torch.cuda.device_count() and not rehearsal:
        raise SystemExit("bench.py: %d ranks but %d GPUs (set HNR_BENCH_REHEARSAL=1 to rehearse the control flow on shared devices)"
                         % (world, torch.cuda.device_count()))
    if rehearsal:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    coll = (lambda t: t.cpu()) if rehearsal else (lambda t: t)
    from hybridneuralrendering_amd import parallel
    from hybridneuralrendering_amd._lib import CNT

    strong = args.scaling == "strong"
    # strong: every rank builds the SAME frame (pose 0) and renders its block of rays; weak: rank-specific pose, whole frame
    sc, opt, agg, cloud, rnd, cam = build_world(args, dev, 0 if strong else rank)
    R_frame = cam["raydir"].shape[0]
    line = sc.w - 2 * args.margin                            # rays per scan line of the frame
    def rays_of(r):                                          # ray indices of rank r (strong scaling)
        if args.shard == "lines":
            return parallel.shard_lines(R_frame, line * max(1, args.band), world, r)
        return torch.arange(*parallel.shard_bounds(R_frame, world, r), dtype=torch.int64)
    shards = [rays_of(r) for r in range(world)] if strong else [torch.arange(R_frame, dtype=torch.int64)] * world
    emulate = os.environ.get("HNR_BENCH_EMULATE_RANK")      # "r/n" on ONE GPU: render only what rank r of n would (tools/predict_scaling.sh)
    if emulate and world == 1:
        er, en = (int(x) for x in emulate.split("/"))
        saved_world, world = world, en
        mine = rays_of(er)
        world = saved_world
        cam = dict(cam, raydir=cam["raydir"].index_select(0, mine.to(dev)).contiguous(), rays_np=cam["rays_np"][mine.numpy()])
        shards = [torch.arange(mine.numel(), dtype=torch.int64)]
    cam_full = cam
    if strong and world > 1:
        mine = shards[rank]
        cam = dict(cam, raydir=cam["raydir"].index_select(0, mine.to(dev)).contiguous(), rays_np=cam["rays_np"][mine.numpy()])
    R = cam["raydir"].shape[0]
    R_job = R_frame if strong else world * R_frame          # rays the whole job renders per step
    if emulate and world == 1:
        R_job = R                                            # the line then describes ONE rank's share, not the frame
    pad = max(int(s.numel()) for s in shards)
    shards_at = [s if rehearsal else s.to(dev) for s in shards] if rank == 0 else None     # where the gathered rows live (rehearsal: host)
    gather_ev = []
    statuses = []                                             # device status words of every launch of the timed loop (read once, after it)

    def step(timers=None, time_gather=False):
        rnd._fm_key = None          # a new frame has new reference views: their feature pyramid is rebuilt inside every step
        if rehearsal and world > 1:
            # rehearsal (all ranks on ONE GPU): the ranks take turns on the device.  Processes that share a GPU are time-sliced by wave
            # preemption, and on this pool a preempted long kernel can resume with a perturbed result (tools/stress_determinism.py,
            # profiles/README.md: 216 of 285 200 pixels of a block); the rehearsal checks the sharding / gather path, not throughput
            for r in range(world):
                if r == rank:
                    col, out = render_frame(rnd, cloud, cam, sc, args.chunk, timers, statuses)
                    torch.cuda.synchronize()
                dist.barrier()
        else:
            col, out = render_frame(rnd, cloud, cam, sc, args.chunk, timers, statuses)
        frame = col
        if world > 1:
            # reassemble the frame (strong) / the N frames (weak) on rank 0: ONE gather over xGMI, equal-size blocks
            buf = col
            if col.shape[0] != pad:
                buf = torch.zeros((pad, 3), dtype=col.dtype, device=col.device)
                buf[:col.shape[0]] = col
            c = coll(buf.contiguous())
            outs = [torch.empty_like(c) for _ in range(world)] if rank == 0 else None
            if time_gather and not rehearsal:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            dist.gather(c, outs, dst=0)
            if time_gather and not rehearsal:
                e1.record()
                gather_ev.append((e0, e1))
            if rank == 0:
                if strong:                                   # every shard's rows go back to their place in the frame
                    frame = torch.empty((R_frame, 3), dtype=c.dtype, device=c.device)
                    for o, s in zip(outs, shards_at):
                        frame[s] = o[:s.numel()]
                else:
                    frame = outs[0]
        return col, out, frame

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    timers = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        col, out, frame = step(timers, time_gather=True)
    barrier()
    dt = time.perf_counter() - t0
    # every rank: an overflow of a single-call workspace (samples dropped) must fail the run, not shade the number
    rnd.check_status(statuses)
    tmine = coll(torch.tensor([dt], dtype=torch.float64, device=dev))
    per_rank = [tmine.clone() for _ in range(world)]
    tmax = tmine.clone()
    if world > 1:
        dist.all_gather(per_rank, tmine)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    per_rank_ms = [round(float(t.item()) / args.steps * 1e3, 3) for t in per_rank]
    dt = float(tmax.item())
    # N > 1, strong scaling: the OTHER way of dealing the frame's rays (scan lines round-robin vs contiguous blocks) in the same run, same steps,
    # so that one record settles the choice (round-4 verdict item 6).  Render only (the gather moves the same bytes either way); max over ranks.
    shard_ab = None
    if strong and world > 1:
        other = "blocks" if args.shard == "lines" else "lines"
        if other == "lines":
            mine_o = parallel.shard_lines(R_frame, line * max(1, args.band), world, rank)
        else:
            mine_o = torch.arange(*parallel.shard_bounds(R_frame, world, rank), dtype=torch.int64)
        cam_o = dict(cam_full, raydir=cam_full["raydir"].index_select(0, mine_o.to(dev)).contiguous(), rays_np=cam_full["rays_np"][mine_o.numpy()])
        st_o = []
        def step_o():
            rnd._fm_key = None
            if rehearsal:
                for r in range(world):
                    if r == rank:
                        render_frame(rnd, cloud, cam_o, sc, args.chunk, None, st_o)
                        torch.cuda.synchronize()
                    dist.barrier()
            else:
                render_frame(rnd, cloud, cam_o, sc, args.chunk, None, st_o)
        step_o()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_o()
        barrier()
        to = coll(torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev))
        per_o = [to.clone() for _ in range(world)]
        dist.all_gather(per_o, to)
        rnd.check_status(st_o)
        ms_o = [round(float(t.item()) / args.steps * 1e3, 3) for t in per_o]
        # (the headline loop's per-rank times include the gather; its render-only counterpart is the stage sum)
        shard_ab = {args.shard: dict(ms_per_step_max_rank=max(per_rank_ms), per_rank_ms=per_rank_ms, includes_gather=True),
                    other: dict(ms_per_step_max_rank=max(ms_o), per_rank_ms=ms_o, includes_gather=False),
                    "note": "same run, same frame, same steps; `%s` is what `value` is quoted on" % args.shard}
    if rank == 0 and args.dump_colors:
        np.save(args.dump_colors, frame.detach().cpu().numpy())
    train_sharded = None
    if not args.no_train_leg:
        train_sharded = train_leg_sharded(args, sc, opt, agg, cloud, rnd, cam, dev, world, rank, rehearsal, emulate)

    if rank == 0:
        counts = out["counts"].cpu().numpy() if args.chunk <= 0 or args.chunk >= R else None
        # stage times inside the timed region: HIP events recorded on the launch stream -- by torch around the Python-driven stages, by
        # the library itself at the stage boundaries of the single-call path (hnr_render_forward's stage_events hook)
        stage_ms = {k: sum(e0.elapsed_time(e1) for e0, e1 in v) / args.steps for k, v in timers.items() if k != "_stage_events"}
        for ev in timers.get("_stage_events", []):
            for k, ms in ev.elapsed_ms().items():
                stage_ms[k] = stage_ms.get(k, 0.0) + ms / args.steps
        # --- roofline of the dominant kernel (fp32 MFMA dense layer) and of the query stage, from HIP events
        # recorded on the launch stream inside the timed region
        roof, roof_q = None, None
        pmc = pmc_traffic() if (int(args.points) == 2000000 and args.scene == "scene0241" and args.chunk <= 0) else {}
        fused = getattr(rnd, "dense", "f32") == "f16x2" and opt.K == 8
        # HNR_DENSE=f32: four linear_f32_kernel launches per frame for the per-neighbour layers
        kname = "linear_f32_kernel<2, 2, 1, 0, 4"
        lin = {k: v for k, v in pmc.items() if kname in k}
        t_lin = None
        if lin:
            n = sum(v["launches"] for v in lin.values())
            t_lin = dict(hbm_bytes=sum(v["hbm_bytes"] * v["launches"] for v in lin.values()) / max(n, 1))
        # (round 5: the k-NN over the grid's 3x3x3 neighbourhood lists: knn_nb_kernel<8, order, 2>)
        knn_name = "knn_nb_kernel<8, 1" if rnd.knn_order == "sorted" else "knn_nb_kernel<8, 0"
        t_q = [v for k, v in pmc.items() if "march_kernel" in k or knn_name in k]
        if counts is not None:
            n_rows, n_valid = int(counts[CNT["NEIGHBOURS"]]), int(counts[CNT["SAMPLES_VALID"]])
            s_all, cells, cand = int(counts[CNT["SAMPLES"]]), int(counts[CNT["CELLS_VISITED"]]), int(counts[CNT["CANDIDATES"]])
            ms_nb = stage_ms.get("mlp_neighbour", 0.0)
            ms_3 = sum(stage_ms.get(k, 0.0) for k in ("dense_b1_2", "dense_b3_0", "dense_b3_2"))
            # ALGORITHMIC flops (271 104 MAC per valid neighbour for block1 + block3)
            flops_nb = 2.0 * n_rows * 256 * (284 + 256 + 263 + 256)
            ms_ch = stage_ms.get("chain", 0.0)
            if fused and ms_ch > 0:
                # dominant kernel: the fused per-neighbour chain (csrc/chain_ws.hip), ONE launch per frame.  Rows are padded to 8 slots
                # per valid sample with more than four neighbours, 4 slots for the others (hnr_chain_plan's two classes), and to whole
                # 128-row tiles per class; every fp32 product is issued as THREE fp16 MFMA products (two-term operand split), K rounded
                # up to 16 per layer (60 -> 64, 263 -> 272).
                n_small, n_tiny = int(counts[CNT["SAMPLES_SMALL"]]), int(counts[CNT["SAMPLES_TINY"]])
                rows_pad = 128 * ((n_valid - n_small - n_tiny + 15) // 16 + (n_small + 31) // 32 + (n_tiny + 63) // 64)
                issued = 3.0 * 2.0 * rows_pad * 256 * (64 + 256 + 272 + 256)
                alg = 2.0 * n_rows * 256 * (60 + 256 + 263 + 256) + 2.0 * n_rows * 256          # executed layers + alpha branch (SURVEY 8d counts 284 columns
                ach = issued / (ms_ch * 1e-3) / 1e12                                             # for block1.0: 224 of them live in the per-point table)
                variant = os.environ.get("HNR_CHAIN_RT", "16")
                kname = {"16": "chain_ws_kernel", "4": "chain_kernel<4"}.get(variant, "chain_ws_kernel")
                t_ch = [v for k, v in pmc.items() if kname in k]
                alg8d = 542720.0 * n_rows                                                        # 2 x 256 x (284 + 256 + 263 + 256) + 2 x 256 flop per valid neighbour
                ach_alg = alg8d / (ms_ch * 1e-3) / 1e12
                roof = dict(kernel="%s: block1 -> block3 -> alpha + K-sums fused (1 launch, %d valid neighbour rows in %d padded rows)" % (
                                {"chain_ws_kernel": "chain_ws_kernel<0> (weight-stationary, epilogue pieces between the wave's own MFMAs)"}.get(kname, kname), n_rows, rows_pad),
                            bound="mfma", achieved=round(ach_alg, 1), peak=BF16_MFMA_PEAK_TF, unit="TFLOP/s", frac=round(ach_alg / BF16_MFMA_PEAK_TF, 4),
                            achieved_issued=round(ach, 1), frac_issued=round(ach / BF16_MFMA_PEAK_TF, 4),
                            frac_executed_fp32=round(alg / (ms_ch * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF, 4),
                            traffic=int(t_ch[0]["hbm_bytes"]) if t_ch else None,
                            traffic_source=("profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)" % TRAFFIC_JSON) if t_ch else None,
                            flops_per_launch=alg8d, flops_issued_per_launch=issued, avg_launch_ms=round(ms_ch, 4),
                            fp32_equivalent_tflops=round(ach_alg, 2),
                            executed_fp32_flops_per_launch=alg,
                            algorithmic_bytes_per_launch=int(n_rows * 168 + n_valid * 1028),
                            note="achieved / frac = ALGORITHMIC flops (SURVEY 8d: 542 720 per valid neighbour, block1.0 counted with all 284 input columns) / HIP-event "
                                 "time of the launch / the 2.5 PFLOP/s dense 16-bit peak.  achieved_issued / frac_issued = 16-bit MFMA flops issued (3 per fp32 product: "
                                 "wm*xh + wh*xm + wh*xh, fp16 two-term split with exact power-of-two row / layer scales, fp32 accumulate; K and row-slot padding) -- what "
                                 "the matrix pipe does, the figure mfma_busy corroborates.  frac_executed_fp32 = 2 M N K of the layers as executed (224 of block1.0's columns "
                                 "live in the per-point table) on the valid rows.  algorithmic bytes = 168 B per valid neighbour (SURVEY 8d) + 1028 B of sums per valid "
                                 "sample; padding to 8 row slots per sample (4 for the %d samples with three or four neighbours, 2 for the %d with one or two) costs "
                                 "%.1f %% extra rows" % (n_small, n_tiny, 100.0 * (rows_pad / max(n_rows, 1) - 1.0)),
                            neighbour_stage=dict(chain_ms=round(ms_ch, 3), gather_ms=round(stage_ms.get("chain_gather", 0.0), 3)))
                try:
                    pm = json.load(open(os.path.join(ROOT, "profiles", CHAIN_PMC_JSON)))
                    roof["mfma_busy"] = pm["chain_ws_kernel"]["mfma_busy_fraction"]
                    roof["mfma_busy_source"] = "profiles/%s (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles from GRBM_GUI_ACTIVE), separate passes)" % CHAIN_PMC_JSON
                except Exception:
                    roof["mfma_busy"] = None
            elif ms_nb > 0:
                # per-neighbour MLP on fp32 MFMA: 4 launches of linear_f32_kernel<2,2,1,0,4>; the kernels EXECUTE fewer flops than
                # the algorithmic count because block1.0's 224 point-only input columns are folded into a per-point table
                flops_exec = 2.0 * n_rows * 256 * (60 + 256 + 263 + 256) if rnd.split_block1 else flops_nb
                ach = flops_nb / (ms_nb * 1e-3) / 1e12
                roof = dict(kernel="linear_f32_kernel<2,2,1,0,4,*> (block1+block3, 4 launches, M=%d rows)" % n_rows, bound="mfma",
                            achieved=round(ach, 2), peak=F32_MFMA_PEAK_TF, unit="TFLOP/s", frac=round(ach / F32_MFMA_PEAK_TF, 4),
                            traffic=int(t_lin["hbm_bytes"]) if t_lin else None,
                            traffic_source=("profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)" % TRAFFIC_JSON) if t_lin else None,
                            flops_per_launch=flops_nb / 4, avg_launch_ms=round(ms_nb / 4, 4),
                            executed_tflops=round(flops_exec / (ms_nb * 1e-3) / 1e12, 2),
                            note="achieved = algorithmic flops / time; executed_tflops = MFMA flops actually issued / time")
            D, K = opt.z_depth_dim, opt.K
            alg = R * (12 + (D + 7) // 8 + 1) + s_all * (12 + 27 * 4 + 4 * K) + 4 * cells + 16 * cand
            ms_q = stage_ms.get("query", 0.0)
            if ms_q > 0:
                ach = alg / (ms_q * 1e-3) / 1e9
                roof_q = dict(kernel="hnr_march_query: march_kernel + worklist scans + k-NN over the neighbourhood lists", bound="hbm", achieved=round(ach, 1),
                              peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                              traffic=int(sum(v["hbm_bytes"] for v in t_q)) if len(t_q) == 2 else None,
                              traffic_source=("profiles/%s (march_kernel + knn kernel, bytes per launch)" % TRAFFIC_JSON) if len(t_q) == 2 else None,
                              algorithmic_bytes=int(alg), avg_launch_ms=round(ms_q, 4),
                              per_ray=dict(samples=round(s_all / R, 2), cells_per_sample=round(cells / max(s_all, 1), 2),
                                           candidates_per_sample=round(cand / max(s_all, 1), 2)))
            # the query in both neighbour orders, timed beside the timed region on the same frame; the roofline is quoted on the order the
            # frame was rendered with (--knn-order)
            if roof_q is not None and opt.K == 8:
                from hybridneuralrendering_amd import querier as Qm
                grid_q, hp_q = rnd.querier._grid_for(cloud.xyz[None])
                tm_q = rnd.querier._tmid_for(float(sc.near), float(sc.far), opt.z_depth_dim, cam["raydir"].shape[0], dev)
                r2_q = np.float32(hp_q[0] ** 2)
                if grid_q is not None and r2_q is not None:
                    ms_o = {}
                    for order in (0, 1):
                        for _ in range(2):
                            Qm.march_query(grid_q, cam["campos"], cam["raydir"], tm_q, opt.SR, opt.K, r2_q, opt.kernel_size, pad=False, knn_order=order)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(5):
                            Qm.march_query(grid_q, cam["campos"], cam["raydir"], tm_q, opt.SR, opt.K, r2_q, opt.kernel_size, pad=False, knn_order=order)
                        e1.record(); torch.cuda.synchronize()
                        ms_o[order] = e0.elapsed_time(e1) / 5
                    # the in-frame query time shares the GPU with the feature-pyramid rebuild on the side stream: the roofline is quoted on the
                    # query alone (same frame, same buffers, 5 launches)
                    fo = 1 if rnd.knn_order == "sorted" else 0
                    roof_q.update(achieved=round(alg / (ms_o[fo] * 1e-3) / 1e9, 1), frac=round(alg / (ms_o[fo] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  avg_launch_ms=round(ms_o[fo], 4), in_frame_ms=round(ms_q, 4),
                                  neighbour_order=("sorted: the reference's neighbour sets in ascending (d2, enumeration) order (hnr_query_params.knn_order = 1)"
                                                   if fo else "reference: slot for slot the reference's insertion history"),
                                  kernel="hnr_march_query: march_kernel + worklist scans + %s (k-NN over the grid's 3x3x3 neighbourhood lists)" % (
                                      "knn_nb_kernel<8,1,2>" if fo else "knn_nb_kernel<8,0,2>"),
                                  timing="HIP events around 5 back-to-back hnr_march_query launches on the bench frame (in the frame the query overlaps the "
                                         "feature-pyramid rebuild on a side stream: in_frame_ms)")
                    roof_q["sorted_neighbour_order"] = dict(avg_launch_ms=round(ms_o[1], 4), reference_order_ms_same_loop=round(ms_o[0], 4),
                                                            achieved=round(alg / (ms_o[1] * 1e-3) / 1e9, 1), frac=round(alg / (ms_o[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                            note="hnr_query_params.knn_order = 1: the reference's neighbour sets in ascending (d2, enumeration) order (v_med3 insertion network); reference "
                                                                 "order: the replay of the reference's farthest-first replacement; both one lane per sample over the grid's neighbourhood "
                                                                 "lists, samples sorted by list length and cell inside a workgroup; counters in profiles/r05_query_pmc.txt")
        # one-off work that is amortised over frames (rebuilt only when the cloud / the weights change), timed once here
        amort = {}
        def _timed(fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            return round(e0.elapsed_time(e1), 3)
        from hybridneuralrendering_amd import querier as Q
        hp = rnd.querier._hp
        amort["grid_build_ms"] = _timed(lambda: Q.VoxelGrid(cloud.xyz, hp[2][:3], hp[5], hp[6], opt.query_size, opt.P, opt.max_o))
        amort["point_table_ms"] = _timed(lambda: agg.point_table(cloud.emb))
        def _records():
            rnd._rec_key = None
            rnd.point_records(cloud)
        amort["point_records_ms"] = _timed(_records)
        # same-run anchor for `dtype: f32`: the identical frame with every per-neighbour layer on fp32 MFMA (HNR_DENSE=f32: v_mfma_f32_32x32x2_f32,
        # per-stage calls), one warm-up + one timed frame, and its largest colour difference from the f16x2 frame of the timed region
        f32_anchor = None
        if world == 1 and fused and not emulate and not getattr(args, "no_f32_anchor", False):
            from hybridneuralrendering_amd.render import HybridRenderer
            old_env = os.environ.get("HNR_DENSE")
            os.environ["HNR_DENSE"] = "f32"
            try:
                rnd32 = HybridRenderer(opt, agg, dev)
                rnd32.knn_order = rnd.knn_order
                render_frame(rnd32, cloud, cam, sc, args.chunk)
                torch.cuda.synchronize(); ta = time.perf_counter()
                col32, _ = render_frame(rnd32, cloud, cam, sc, args.chunk)
                torch.cuda.synchronize()
                f32_anchor = dict(fp32_mfma_ms_per_step=round((time.perf_counter() - ta) * 1e3, 3),
                                  max_abs_vs_f16x2_frame=float((col32 - col).abs().max()),
                                  note="HNR_DENSE=f32: the per-neighbour layers as four fp32-MFMA launches (the round-1 path, parity-tested); same frame, same process")
                del rnd32, col32
            finally:
                if old_env is None: os.environ.pop("HNR_DENSE", None)
                else: os.environ["HNR_DENSE"] = old_env
        # (the training leg runs BEFORE the CPU baseline: 128 host threads that have just been spinning would perturb a leg whose launches are host-driven)
        train = None
        if world == 1 and not args.no_train_leg and not args.train_sharded_only:
            try:                                                               # a leg reported beside the headline must not take the line down
                train = train_leg(args, sc, opt, agg, cloud, rnd, cam, dev)
            except Exception as ex:                                            # noqa: BLE001
                if os.environ.get("HNR_BENCH_STRICT", "0") == "1": raise
                train = dict(workload="C3 train step", error="%s: %s" % (type(ex).__name__, str(ex)[:300]))
        cpu = None
        if not args.no_cpu_baseline and world == 1:          # reported at N=1 only (rank 0)
            cpu = cpu_baseline(args, sc, opt, agg, cam, col.cpu().numpy())
        res = {
            "metric": "rays/sec (fwd render) scene0241_01 at 1/2/4/8 GPU; PSNR delta vs ref",
            "value": R_job * args.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "dense_arithmetic": {"f32": "fp32 MFMA (v_mfma_f32_32x32x2_f32)", "f16x2": "per-neighbour chain fused in one kernel: fp32 operands split into 2 fp16 terms under exact power-of-two row / layer scales, 3 fp16 MFMAs per product, fp32 accumulate (error vs fp64 at or below the fp32-MFMA path's, tests/test_chain_gpu.py); all other layers fp32 MFMA"}[getattr(rnd, "dense", "f32")], "data": ("synthetic (EMULATION of rank %s on one GPU: not the frame metric)" % emulate) if (emulate and world == 1) else "synthetic" if not rehearsal else "synthetic (REHEARSAL: ranks share GPUs, gloo collectives -- not a measurement)",
            "config": {"workload": "%s synthetic scene (SURVEY 8d): %d points, %dx%d frame margin %d = %d rays per step (%s), "
                                   "SR=%d K=%d P=%d max_o=%d D=%d, 4 reference views %dx%d, hybrid viewmlp forward (query+gather+aggregate+composite), neighbour lists in %s order; "
                                   "random-init weights with alpha_branch.0 rescaled (weight x30, bias = 30) so that opacities spread over (0,1)"
                                   % ({"scene0241": "scene0241_01-like room", "scene0101": "scene0101_04-like room"}.get(args.scene, args.scene + "-like object"),
                                      sc.xyz.shape[0], sc.w, sc.h, args.margin, R_frame,
                                      "ONE fixed frame sharded over the ranks" if strong else "one such frame per rank",
                                      opt.SR, opt.K, opt.P, opt.max_o, opt.z_depth_dim, sc.h, sc.w, rnd.knn_order),
                       "entry": "hnr_render_forward (one library call per frame, no host read)" if getattr(rnd, "single_call", False) and fused else "per-stage C-ABI calls from Python",
                       "rays_per_step": R_job, "rays_per_gpu": R, "points": int(sc.xyz.shape[0]), "chunk_rays": args.chunk if args.chunk > 0 else R,
                       "parallelism": (("one fixed frame ray-sharded x%%d (%s), one RCCL gather" % ("scan lines dealt round-robin" if args.shard == "lines" else "contiguous scan-line blocks")) if strong else
                                       "one frame per rank x%d, one RCCL gather") % world},
            "gather_ms": (round(sum(a.elapsed_time(b) for a, b in gather_ev) / max(len(gather_ev), 1), 4) if gather_ev else None),
            "per_rank_ms_per_step": per_rank_ms, "status_words_checked": len(statuses), "shard_ab": shard_ab,
            "rccl_ranks": (world if (world > 1 and not rehearsal) else 0),
            "fp32_mfma_anchor": f32_anchor,
            "roofline": roof, "roofline_query": roof_q, "roofline_train": (train or {}).get("roofline_train"), "cpu_baseline": cpu,
            "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
            "amortised_ms": amort, "train_step": train, "train_step_sharded": train_sharded, "grid": rnd.querier.last_grid_stats,
        }
        if counts is not None:
            res["counts"] = {k: int(counts[v]) for k, v in CNT.items()}
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
