"""Training legs: BASELINE config C3 (one GPU, fwd + bwd with the shipped loss) and C5 (patch-sharded step with the blur
module)."""
import os
import time

import numpy as np
import torch

from .common import HBM_PEAK_GBS, TRAIN_TRAFFIC_JSON, pmc_traffic


def train_leg_sharded(args, sc, opt, agg, cloud, rnd, cam, dev, world, rank, rehearsal, emulate, steps=20, warmup=3):
    """BASELINE config C5 the way it runs on N GPUs (models/mvs_points_volumetric_model.py:111-152,
    models/base_rendering_model.py:677-745): one batch of
    49 dilated 8x8 patches (3136 rays, dilation_setup 7_8_1_6), the blur-handling module (12 symmetric 9x9 kernels) and
    the item's frame weight; the batch is
    sharded by WHOLE patches (parallel.shard_patches), every rank runs forward -> blur module -> loss kernels -> blur
    backward -> backward on its 6-7 patches
    (train.train_step: no autograd graph, no torch.unique, no host read; HNR_BENCH_TRAIN_GRAPH=1: captured in a hipGraph
    and replayed), then the gradients
    meet in TWO collectives without a host read: ONE all-reduce of the flat weight-gradient buffer carrying the ranks'
    valid-ray counts
    (parallel.allreduce_weight_grads: the loss is a mean over the batch's valid rays) and ONE fixed-capacity all-gather
    of packed (point id | 39 floats)
    records of the touched points (parallel.PointGradExchange; the touched list is what the forward call left on the
    device).
    Timed with HIP events per part; max over ranks.  world == 1 and HNR_BENCH_EMULATE_RANK=r/n: rank r's share of an
    n-way split alone on this GPU -- the
    collectives degenerate to their local pack / apply parts, which are still run and timed
    (tools/predict_train_scaling.sh)."""
    import torch.distributed as dist
    from hybridneuralrendering_amd import scenes, parallel
    from hybridneuralrendering_amd.train import TrainPath, train_step, CapturedTrainStep
    old_train, old_dil = opt.is_train, getattr(opt, "dilation_setup", None)
    opt.is_train, opt.dilation_setup = 1, "7_8_1_6"
    # HNR_BENCH_TRAIN_GRAPH=1: replay the step from a hipGraph (train.CapturedTrainStep).  Measured in round 5 and NOT
    # the default: the ROCm 7.2 graph executor runs the step's three queues one after the other (a 1/8 share: 2.60 ms
    # replayed = the single-queue eager step, 2.18 ms eager with the side streams)
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
        # one rank: the batch in its own (row-major) order
        ray_ids = (torch.arange(S * S) if n_way == 1 else ray_ids).to(dev)
        layout, n_patches = ("grid", pn) if n_way == 1 else ("patch_major", int(ids.numel()))
        path = TrainPath(rnd)
        # a training loop: every step writes the same output / gradient tensors
        path.reuse_outputs = True
        leaves = [x.clone().requires_grad_(True) for x in (cloud.emb, cloud.conf, cloud.dir, cloud.color)]
        for prm in agg.parameters():
            prm.requires_grad_(True)
        my_rays, my_gt, my_drop = rays_all[ray_ids].contiguous(), gt[ray_ids].contiguous(), drop[ray_ids].contiguous()
        w2c = torch.inverse(cam["c2w_nearest"]).contiguous()
        ev = lambda: torch.cuda.Event(enable_timing=True)
        state = dict(cap=None, ex=None)

        def compute():
            """one rank's step: eager (train_step) or a replay of the captured graph; jittered depth tables are drawn
            inside either way"""
            if state["cap"] is not None:
                return state["cap"].step(assign_grads=False)
            return train_step(path, agg, cloud.xyz, leaves[0], leaves[1], leaves[2], leaves[3], my_rays, cam["campos"],
                              cam["camrot"], cam["bg"], sc.near, sc.far,
                              cam["c2w_nearest"], cam["campos_nearest"], cam["intrinsic"], cam["images"], my_gt,
                              zero_epsilon=1e-3, w_color=1.0, w_zero_one=1e-4,
                              frame_weight=frame_weight, ray_drop=my_drop, assign_grads=False, blur_kernels=kern,
                              patch_num=n_patches, patch_size=ps,
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
                    # any step of the loop (the jitter changes the touched set)
                    state["over"] = over if "over" not in state else torch.maximum(state["over"], over)
            elif e:
                e[2].record()
            if e: e[3].record()
            if timed is not None: timed.append(e)
            return out
        # Preflight: this leg is reported BESIDE the headline line, so it must not be able to take the run down or leave
        # ranks waiting in a collective for one that raised.  Every rank runs one eager step (and, by default, captures
        # the step in a hipGraph) without the collectives, the ranks agree on the outcome and on the exchange capacity
        # (one all-reduce that every rank reaches), and only then the collectives run.
        err, n_touched, graph_note = None, 0, "eager: train.train_step, launches queued back to back on three queues"
        try:
            out = one(collect=False)
            torch.cuda.synchronize()
            TrainPath.check_status(out)
            n_touched = int(TrainPath.touched_points(out["_saved"])[1].item())
            if use_graph:
                try:
                    sample = dict(raydir=my_rays, campos=cam["campos"], camrot=cam["camrot"], bg_color=cam["bg"],
                                  c2w_nearest=cam["c2w_nearest"], w2c_nearest=w2c,
                                  campos_nearest=cam["campos_nearest"], intrinsic_nearest=cam["intrinsic"],
                                  images_nearest=cam["images"], gt_image=my_gt,
                                  ray_drop=my_drop, blur_kernels=kern, frame_weight=frame_weight)
                    state["cap"] = CapturedTrainStep(path, agg, cloud.xyz, leaves[0], leaves[1], leaves[2], leaves[3],
                                                     sample, sc.near, sc.far, zero_epsilon=1e-3,
                                                     w_color=1.0, w_zero_one=1e-4, patch_num=n_patches, patch_size=ps,
                                                     patch_layout=layout)
                    graph_note = "hipGraph replay (train.CapturedTrainStep)"
                # noqa: BLE001  (the eager step is the fallback of the MEASUREMENT, not of the product)
                except Exception as ex:
                    if os.environ.get("HNR_BENCH_STRICT"):
                        raise
                    state["cap"] = None
                    graph_note = "eager: capture failed (%s: %s)" % (type(ex).__name__, str(ex)[:200])
        except Exception as ex:                                                  # noqa: BLE001
            if os.environ.get("HNR_BENCH_STRICT"):
                raise
            err = "%s: %s" % (type(ex).__name__, str(ex)[:300])
        flag = torch.tensor([0 if err else 1, -n_touched], dtype=torch.int64,
                            device="cpu" if (rehearsal or world == 1) else dev)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag[0].item()) == 0 and err is None:
                err = "another rank failed its preflight step"
        if err:
            return dict(workload="C5 sharded train step", error=err, n_ranks=n_way)
        # 2 x the busiest rank's touched points of the preflight step
        capacity = max(1024, (int(-flag[1].item()) * 2 + 255) // 256 * 256)
        # (one rank, nothing emulated: there is nothing to exchange)
        state["ex"] = parallel.PointGradExchange(capacity) if n_way > 1 else None
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
            # (N > 1: the other ranks wait in a collective -- fail the run rather than hang it)
            if os.environ.get("HNR_BENCH_STRICT") or world > 1:
                raise
            return dict(workload="C5 sharded train step", error="%s: %s" % (type(ex).__name__, str(ex)[:300]),
                        n_ranks=n_way)
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
            # a rank touched more points than the agreed capacity in some step: its extra rows stayed local, the
            # replicas would diverge -- not a valid timing
            return dict(workload="C5 sharded train step", n_ranks=n_way, exchange_capacity=capacity,
                        error="PointGradExchange overflowed its capacity of %d records in at least one timed step" %
                        capacity)
        return dict(workload="C5: %d dilated %dx%d patches (dilation_setup 7_8_1_6) = %d rays, blur module (12 kernels "
                             "9x9) + frame weight, fwd + bwd%s" % (
                        pn * pn, ps, ps, S * S, "" if n_way == 1 else "; rank %d of %d: %d patches = %d rays" % (r_of,
                                n_way, int(ids.numel()), int(ray_ids.numel()))),
                    ms_per_step=round(dt * 1e3, 3), compute_ms=round(comp * 1e3, 3),
                    allreduce_weights_ms=round(ar_w * 1e3, 3), exchange_points_ms=round(ar_p * 1e3, 3),
                    per_rank_ms_per_step=per_rank, n_ranks=n_way,
                    rccl_ranks=(world if (world > 1 and not rehearsal) else 0), steps=steps, step_form=graph_note,
                    emulated_rank=("%d/%d on one GPU: the collectives are their local pack / apply parts only" % (r_of,
                            n_way)) if (emulate and world == 1) else None,
                    valid_samples=int(c[6]), neighbour_rows=int(c[3]), touched_points=n_touched,
                    exchange_capacity=capacity, exchange_overflow=bool(over),
                    collective_bytes=dict(weights_allreduce=4 * (n_w + 1),
                                          point_records_allgather_per_rank=(capacity + 2) * 40 * 4,
                                          dense_point_allreduce_avoided=int(sum(x.numel() for x in leaves) * 4)),
                    note="max over ranks; no host read in the step; collectives: parallel.allreduce_weight_grads (ONE "
                         "all-reduce of the flat weight-gradient buffer + the "
                         "valid-ray count) and parallel.PointGradExchange (ONE fixed-capacity all-gather of packed (id "
                         "| 39 floats) records, applied in rank order)")
    finally:
        opt.is_train, opt.dilation_setup = old_train, old_dil
        for prm in agg.parameters():
            prm.requires_grad_(False)


def train_leg(args, sc, opt, agg, cloud, rnd, cam, dev, steps=20, warmup=3):
    """SURVEY 8d config C3 (fwd+bwd): one 56x56 = 3136-ray training batch (random window, jittered depths, patch drop)
    through
    the HIP forward + backward with the shipped loss terms.  Reported beside the headline metric, never part of
    `value`."""
    from hybridneuralrendering_amd import scenes
    from hybridneuralrendering_amd.train import TrainPath, train_step
    old = opt.is_train
    opt.is_train = 1
    try:
        path = TrainPath(rnd)
        # a training loop: every step writes the same output / gradient tensors
        path.reuse_outputs = True
        rng = np.random.default_rng(17)
        x0 = int(rng.integers(args.margin, sc.w - args.margin - 56)); y0 = int(rng.integers(args.margin,
                sc.h - args.margin - 56))
        px, py = np.meshgrid(np.arange(x0, x0 + 56), np.arange(y0, y0 + 56), indexing="ij")
        pix = np.stack([px, py], axis=-1).reshape(-1, 2).astype(np.int32)
        raydir = torch.from_numpy(scenes.camera_rays(pix, sc.intrinsic, sc.c2w)).to(dev)
        gt = torch.rand((raydir.shape[0], 3), device=dev)
        # the item's reference-view poses inverted once per item (four 4x4 matrices: data-loader work)
        w2c_c3 = torch.inverse(cam["c2w_nearest"]).contiguous()
        leaves = [t.clone().requires_grad_(True) for t in (cloud.emb, cloud.conf, cloud.dir, cloud.color)]
        for prm in agg.parameters():
            prm.requires_grad_(True)
        def one(ev=None):
            for t in leaves:
                t.grad = None
            agg.zero_grad(set_to_none=True)
            if ev: ev[0].record()
            # forward -> the shipped loss terms (masked colour MSE + zero-one regulariser on conf_coefficient of the
            # valid rays; value and gradients on the device, hnr_shipped_loss_rows) -> backward, queued back to back
            # (train.train_step): no autograd graph, no host read
            out, _pg, _ag = train_step(path, agg, cloud.xyz, leaves[0], leaves[1], leaves[2], leaves[3], raydir,
                                       cam["campos"], cam["camrot"],
                                       cam["bg"], sc.near, sc.far, cam["c2w_nearest"], cam["campos_nearest"],
                                       cam["intrinsic"], cam["images"], gt,
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
        host_dt = (time.perf_counter() - t0) / steps          # the host's share: Python + ctypes + ~200 launches, no wait
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        gpu_ms = sum(e[0].elapsed_time(e[1]) for e in evs) / steps
        c = out["counts"].cpu().numpy()
        # the same step captured once in a hipGraph and replayed (train.CapturedTrainStep; the depth jitter is drawn
        # inside the graph), reported beside the eager number: on ROCm 7.2 the graph executor serialises the step's
        # three queues, so the replay is the SLOWER form
        captured_ms, graph_note = None, "not measured (HNR_BENCH_TRAIN_GRAPH=0)"
        if os.environ.get("HNR_BENCH_TRAIN_GRAPH", "1") != "0":
            try:
                from hybridneuralrendering_amd.train import CapturedTrainStep
                sample = dict(raydir=raydir, campos=cam["campos"], camrot=cam["camrot"], bg_color=cam["bg"],
                              c2w_nearest=cam["c2w_nearest"],
                              campos_nearest=cam["campos_nearest"], intrinsic_nearest=cam["intrinsic"],
                              images_nearest=cam["images"], gt_image=gt)
                capt = CapturedTrainStep(path, agg, cloud.xyz, leaves[0], leaves[1], leaves[2], leaves[3], sample,
                                         sc.near, sc.far, zero_epsilon=1e-3,
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
        # stage times of the two library calls (HIP events recorded by the library at its stage boundaries, one extra
        # step)
        path.timers = {}
        one()
        torch.cuda.synchronize()
        stage = {("fwd." + k): round(v, 4) for k, v in path.timers["fwd"][0].elapsed_ms().items()}
        stage.update({("bwd." + k): round(v, 4) for k, v in path.timers["bwd"][0].elapsed_ms().items()})
        path.timers = None
        # the forward call's share (its stage events); the rest: loss kernels + backward
        fwd = sum(v for k, v in stage.items() if k.startswith("fwd."))
        bwd = gpu_ms - fwd
        # roofline of the step's dominant kernel: the weight-gradient GEMM dW = dZ^T X of a 256 x 256 per-neighbour
        # layer (hnr_h2wgrad, five such launches per step), timed alone with HIP events on tensors of the step's row
        # count (8 row slots per valid sample)
        from hybridneuralrendering_amd import _lib
        Lh = _lib.lib()
        M8 = 8 * int(c[6])
        Zt, Xt = torch.randn((max(M8, 1), 256), device=dev), torch.randn((max(M8, 1), 256), device=dev)
        mz = torch.tensor([np.float32(8.0).view(np.int32)], dtype=torch.int32, device=dev)
        scr = torch.empty((int(Lh.hnr_h2wgrad_scratch_bytes(256, 256)),), dtype=torch.uint8, device=dev)
        dW, db = torch.empty((256, 256), device=dev), torch.empty((256,), device=dev)
        def wg():
            _lib.check(Lh.hnr_h2wgrad(_lib.ptr(Zt), 256, _lib.ptr(Xt), 256, M8, None, 1, 0, 256, 256, _lib.ptr(mz),
                                      _lib.ptr(mz), _lib.ptr(dW), 256, _lib.ptr(db), 0,
                                      _lib.ptr(scr), _lib.stream()), "hnr_h2wgrad")
        for _ in range(3):
            wg()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            wg()
        e1.record(); torch.cuda.synchronize()
        ms_w = e0.elapsed_time(e1) / 10
        # 3 fp16 MFMAs per fp32 product, K + 1 (bias column) padded to 9 tiles of 32
        issued = 3.0 * 2.0 * M8 * 256 * 288
        # HBM bytes of the same kernel inside the step (PMC passes over tools/probe_train.py; only valid for the default
        # C3 batch: 307 120 row slots) PMC bytes of the 256-wide weight gradient (its launches inside the training step,
        # profiles/<TRAIN_TRAFFIC_JSON>), selected by kernel name; the batch behind that file is this one up to the
        # depth jitter (row slots within 1 %: `traffic_rows` beside it)
        # (the DMA-staged kernel specialised for the 256-wide layers: the default since round 6)
        t_wg = [v for k, v in pmc_traffic(TRAIN_TRAFFIC_JSON).items() if "h2wgrad_dma256_kernel<false>" in k]
        roof_t = dict(kernel="h2wgrad_dma256_kernel<false> + reduce (hnr_h2wgrad: dW = dZ^T X, db of one 256 x 256 per-neighbour "
                             "layer; M = %d row slots)" % M8, bound="hbm",
                      achieved=round(M8 * 2048.0 / (ms_w * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                      frac=round(M8 * 2048.0 / (ms_w * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      traffic=int(t_wg[0]["hbm_bytes"]) if t_wg else None,
                      traffic_source=("profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE over "
                                      "tools/probe_train.py, bytes per launch: operands + %d KiB of "
                                      "per-workgroup partial sums)" % (TRAIN_TRAFFIC_JSON,
                                              256 * 288 * 4 * 256 // 1024)) if t_wg else None,
                      avg_launch_ms=round(ms_w, 4), algorithmic_bytes_per_launch=int(M8 * 2048),
                      mfma_tflops_issued=round(issued / (ms_w * 1e-3) / 1e12, 1),
                      fp32_equivalent_tflops=round(2.0 * M8 * 256 * 256 / (ms_w * 1e-3) / 1e12, 1),
                      note="algorithmic bytes = the two fp32 operands read once (2 KiB per row); the f16x2 MFMA work "
                           "of this shape (3 x 2 M N K) would take "
                           "%.3f ms at the 2.5 PFLOP/s peak, the operand stream %.3f ms at 8 TB/s: HBM is the nearer "
                           "roof" % (issued / 2.5e15 * 1e3, M8 * 2048.0 / 8e12 * 1e3)) if M8 > 0 else None
        return dict(workload="C3: 56x56 = %d rays, fwd (train mode) + bwd, shipped loss" % raydir.shape[0],
                    ms_per_step=round(dt * 1e3, 3), host_ms_per_step=round(host_dt * 1e3, 3),
                    captured_ms_per_step=captured_ms, captured_form=graph_note, rays_per_s=round(raydir.shape[0] / dt,
                            1), fwd_ms=round(fwd, 3), loss_bwd_ms=round(bwd, 3),
                    neighbour_rows=int(c[3]), valid_samples=int(c[6]), steps=steps, entry="hnr_render_train_forward + "
                                                                                          "hnr_render_train_backward "
                                                                                          "(two library calls per "
                                                                                          "step, no host read)",
                    stage_ms=stage, roofline_train=roof_t)
    finally:
        opt.is_train = old
        for prm in agg.parameters():
            prm.requires_grad_(False)

