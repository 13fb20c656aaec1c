"""The timed frame loop, the rooflines and the JSON line (rank 0 prints it)."""
import json
import os
import time

import numpy as np
import torch

from . import rooflines as R_
from .common import parse, spawn_ranks, build_world, render_frame
from .cpu_leg import cpu_baseline
from .train_legs import train_leg, train_leg_sharded


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # the parent has not touched the GPU (device_count() / is_available() not called yet)
        return spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher set WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # Rehearsal of the multi-rank control flow on a box with fewer GPUs than ranks (HNR_BENCH_REHEARSAL=1 only: ranks
    # share devices and the collectives run over gloo on host copies -- numbers from such a run mean nothing and say
    # so).
    rehearsal = world > 1 and os.environ.get("HNR_BENCH_REHEARSAL") == "1" and torch.cuda.device_count() < world
    if world > torch.cuda.device_count() and not rehearsal:
        raise SystemExit("bench.py: %d ranks but %d GPUs (set HNR_BENCH_REHEARSAL=1 to rehearse the control flow on "
                         "shared devices)"
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
    # strong: every rank builds the SAME frame (pose 0) and renders its block of rays; weak: rank-specific pose, whole
    # frame
    sc, opt, agg, cloud, rnd, cam = build_world(args, dev, 0 if strong else rank)
    R_frame = cam["raydir"].shape[0]
    line = sc.w - 2 * args.margin                            # rays per scan line of the frame
    def rays_of(r):                                          # ray indices of rank r (strong scaling)
        if args.shard == "lines":
            return parallel.shard_lines(R_frame, line * max(1, args.band), world, r)
        return torch.arange(*parallel.shard_bounds(R_frame, world, r), dtype=torch.int64)
    shards = [rays_of(r) for r in range(world)] if strong else [torch.arange(R_frame, dtype=torch.int64)] * world
    # "r/n" on ONE GPU: render only what rank r of n would (tools/predict_scaling.sh)
    emulate = os.environ.get("HNR_BENCH_EMULATE_RANK")
    if emulate and world == 1:
        er, en = (int(x) for x in emulate.split("/"))
        saved_world, world = world, en
        mine = rays_of(er)
        world = saved_world
        cam = dict(cam, raydir=cam["raydir"].index_select(0, mine.to(dev)).contiguous(),
                   rays_np=cam["rays_np"][mine.numpy()])
        shards = [torch.arange(mine.numel(), dtype=torch.int64)]
    cam_full = cam
    if strong and world > 1:
        mine = shards[rank]
        cam = dict(cam, raydir=cam["raydir"].index_select(0, mine.to(dev)).contiguous(),
                   rays_np=cam["rays_np"][mine.numpy()])
    R = cam["raydir"].shape[0]
    R_job = R_frame if strong else world * R_frame          # rays the whole job renders per step
    if emulate and world == 1:
        R_job = R                                            # the line then describes ONE rank's share, not the frame
    pad = max(int(s.numel()) for s in shards)
    # where the gathered rows live (rehearsal: host)
    shards_at = [s if rehearsal else s.to(dev) for s in shards] if rank == 0 else None
    gather_ev = []
    # device status words of every launch of the timed loop (read once, after it)
    statuses = []

    def step(timers=None, time_gather=False):
        # a new frame has new reference views: their feature pyramid is rebuilt inside every step
        rnd._fm_key = None
        if rehearsal and world > 1:
            # rehearsal (all ranks on ONE GPU): the ranks take turns on the device.  Processes that share a GPU are
            # time-sliced by wave preemption, and on this pool a preempted long kernel can resume with a perturbed
            # result (tools/stress_determinism.py, profiles/README.md: 216 of 285 200 pixels of a block); the rehearsal
            # checks the sharding / gather path, not throughput
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
    # N > 1, strong scaling: the OTHER way of dealing the frame's rays (scan lines round-robin vs contiguous blocks,
    # SURVEY 8e) in the same run, same steps, so that one record settles the choice (round-4 verdict item 6).  Render
    # only (the gather moves the same bytes either way); max over ranks.
    shard_ab = None
    if strong and world > 1:
        other = "blocks" if args.shard == "lines" else "lines"
        if other == "lines":
            mine_o = parallel.shard_lines(R_frame, line * max(1, args.band), world, rank)
        else:
            mine_o = torch.arange(*parallel.shard_bounds(R_frame, world, rank), dtype=torch.int64)
        cam_o = dict(cam_full, raydir=cam_full["raydir"].index_select(0, mine_o.to(dev)).contiguous(),
                     rays_np=cam_full["rays_np"][mine_o.numpy()])
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
        shard_ab = {args.shard: dict(ms_per_step_max_rank=max(per_rank_ms), per_rank_ms=per_rank_ms,
                                     includes_gather=True),
                    other: dict(ms_per_step_max_rank=max(ms_o), per_rank_ms=ms_o, includes_gather=False),
                    "note": "same run, same frame, same steps; `%s` is what `value` is quoted on" % args.shard}
    if rank == 0 and args.dump_colors:
        np.save(args.dump_colors, frame.detach().cpu().numpy())
    train_sharded = None
    if not args.no_train_leg:
        train_sharded = train_leg_sharded(args, sc, opt, agg, cloud, rnd, cam, dev, world, rank, rehearsal, emulate)
        if world == 1 and not emulate and train_sharded and not train_sharded.get("error"):
            # what sharding this batch over 8 GPUs could buy (round-5 verdict): rank 0's 1/8 share alone on this GPU,
            # the collectives' local parts included; per-GPU efficiency = whole batch / (8 x share).  C5 is too small a
            # batch to shard efficiently and the line says so.
            share = train_leg_sharded(args, sc, opt, agg, cloud, rnd, cam, dev, world, rank, rehearsal, "0/8", steps=10)
            if share and not share.get("error") and share.get("ms_per_step"):
                eff = train_sharded["ms_per_step"] / (8.0 * share["ms_per_step"])
                train_sharded["share_1_of_8"] = dict(
                    ms_per_step=share["ms_per_step"], compute_ms=share.get("compute_ms"),
                    predicted_per_gpu_efficiency_at_8=round(eff, 3),
                    note="emulated on ONE GPU (no RCCL): one rank's 6-7 patches alone; a share is a chain of ~190 "
                         "launches whose fixed part does not shrink with the batch")

    if rank == 0:
        counts = out["counts"].cpu().numpy() if args.chunk <= 0 or args.chunk >= R else None
        stage_ms = R_.stage_times(timers, args.steps)
        fused = getattr(rnd, "dense", "f32") == "f16x2" and opt.K == 8
        roof = roof_q = None
        if counts is not None:
            roof = R_.chain_roofline(args, rnd, opt, counts, stage_ms, CNT)
            roof_q = R_.query_roofline(args, rnd, opt, cloud, sc, cam, dev, counts, stage_ms, CNT)
        amort = R_.amortised(rnd, agg, cloud, opt)
        resident = R_.resident_bytes(rnd, agg, cloud, cam, opt)
        f32_anchor = None
        if world == 1 and fused and not emulate and not getattr(args, "no_f32_anchor", False):
            f32_anchor = R_.fp32_anchor(args, rnd, opt, agg, cloud, cam, sc, dev, col)
        # (the training leg runs BEFORE the CPU baseline: 128 host threads that have just been spinning would perturb a
        # leg whose launches are host-driven)
        train = None
        if world == 1 and not args.no_train_leg and not args.train_sharded_only:
            try:                            # a leg reported beside the headline must not take the line down
                train = train_leg(args, sc, opt, agg, cloud, rnd, cam, dev)
            except Exception as ex:         # noqa: BLE001
                if os.environ.get("HNR_BENCH_STRICT", "0") == "1":
                    raise
                train = dict(workload="C3 train step", error="%s: %s" % (type(ex).__name__, str(ex)[:300]))
        cpu = None
        if not args.no_cpu_baseline and world == 1:          # reported at N=1 only (rank 0)
            cpu = cpu_baseline(args, sc, opt, agg, cam, col.cpu().numpy())
        data = "synthetic"
        if emulate and world == 1:
            data = "synthetic (EMULATION of rank %s on one GPU: not the frame metric)" % emulate
        elif rehearsal:
            data = "synthetic (REHEARSAL: ranks share GPUs, gloo collectives -- not a measurement)"
        res = {
            "metric": "rays/sec (fwd render) scene0241_01 at 1/2/4/8 GPU; PSNR delta vs ref",
            "value": R_job * args.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32", "dense_arithmetic": R_.DENSE_NOTE[getattr(rnd, "dense", "f32")], "data": data,
            "config": R_.describe_config(args, sc, opt, rnd, R_frame, R_job, R, world, strong),
            "gather_ms": (round(sum(a.elapsed_time(b) for a, b in gather_ev) / max(len(gather_ev), 1),
                                4) if gather_ev else None),
            "per_rank_ms_per_step": per_rank_ms, "status_words_checked": len(statuses), "shard_ab": shard_ab,
            "rccl_ranks": (world if (world > 1 and not rehearsal) else 0),
            "fp32_mfma_anchor": f32_anchor,
            "roofline": roof, "roofline_query": roof_q, "roofline_train": (train or {}).get("roofline_train"),
            "cpu_baseline": cpu,
            "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
            "amortised_ms": amort, "resident_bytes": resident, "train_step": train, "train_step_sharded": train_sharded,
            "grid": rnd.querier.last_grid_stats,
        }
        if counts is not None:
            res["counts"] = {k: int(counts[v]) for k, v in CNT.items()}
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()

