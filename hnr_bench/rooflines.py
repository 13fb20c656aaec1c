"""Roofline objects of the bench line and the one-off measurements reported beside it (rank 0, after the timed loop).

roofline        = the dominant kernel (chain_ws_kernel, MFMA-bound): ALGORITHMIC flops (542 720 per valid
                  neighbour) / the launch's HIP-event time / the 2.5 PFLOP/s dense 16-bit peak;
roofline_query  = the point-query stage (march + work list + k-NN, HBM-bound by SURVEY 8d's accounting): algorithmic
                  bytes counted from the query's own device counters / HIP-event time of the stage alone / 8 TB/s.
`traffic` / `mfma_busy` come from separate rocprofv3 --pmc passes kept under profiles/ (tools/gpu_job.sh traffic|pmc); a
PMC file that was collected from ANOTHER build of the kernel source is refused (source_sha256 recorded by
tools/collect_pmc.py)."""
import hashlib
import json
import os
import time

import numpy as np
import torch

from .common import (ROOT, HBM_PEAK_GBS, F32_MFMA_PEAK_TF, BF16_MFMA_PEAK_TF, TRAFFIC_JSON, CHAIN_PMC_JSON, pmc_traffic,
                     render_frame)

CSRC = os.path.join(ROOT, "hybridneuralrendering_amd", "csrc")
PMC_NOTE = "rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch"


def source_sha256(name):
    """sha256 of a kernel source file: the identity of the code a PMC file describes (the GPU box has no .git)."""
    try:
        return hashlib.sha256(open(os.path.join(CSRC, name), "rb").read()).hexdigest()
    except OSError:
        return None


def stage_times(timers, steps):
    """ms per step of every stage inside the timed region: HIP events recorded on the launch stream -- by torch around
    the Python-driven stages, by the library itself at the stage boundaries of the single call (stage_events hook)."""
    stage_ms = {k: sum(e0.elapsed_time(e1) for e0, e1 in v) / steps for k, v in timers.items() if k != "_stage_events"}
    for ev in timers.get("_stage_events", []):
        for k, ms in ev.elapsed_ms().items():
            stage_ms[k] = stage_ms.get(k, 0.0) + ms / steps
    return stage_ms


def profiled_workload(args):
    """The PMC files under profiles/ describe the default workload only."""
    return int(args.points) == 2000000 and args.scene == "scene0241" and args.chunk <= 0


def chain_mfma_busy():
    """(mfma_busy, source) from profiles/<round>_chain_pmc.json, or (None, why) when the file is missing or stale."""
    path = os.path.join(ROOT, "profiles", CHAIN_PMC_JSON)
    try:
        pm = json.load(open(path))
    except Exception:
        return None, "no profiles/%s" % CHAIN_PMC_JSON
    want, have = source_sha256("chain_ws.hip"), (pm.get("source_sha256") or {}).get("chain_ws.hip")
    if have is None or want is None or have != want:
        return None, ("profiles/%s was collected from another build of csrc/chain_ws.hip (sha256 %s, now %s): "
                      "re-collect it "
                      "(tools/gpu_job.sh pmc)" % (CHAIN_PMC_JSON, str(have)[:12], str(want)[:12]))
    return pm["chain_ws_kernel"]["mfma_busy_fraction"], ("profiles/%s: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel "
                                                         "cycles "
                                                         "from GRBM_GUI_ACTIVE), separate passes" % CHAIN_PMC_JSON)


def chain_roofline(args, rnd, opt, counts, stage_ms, CNT):
    """Dominant kernel.  f16x2 (default): the fused per-neighbour chain, ONE launch per frame; rows are padded to 8 / 4
    / 2 slots per valid sample (hnr_chain_plan's classes) and to whole 128-row tiles; every fp32 product is issued as
    THREE fp16 MFMA products, K rounded up to 16 per layer (60 -> 64, 263 -> 272).  HNR_DENSE=f32: four fp32-MFMA
    launches."""
    pmc = pmc_traffic() if profiled_workload(args) else {}
    n_rows, n_valid = int(counts[CNT["NEIGHBOURS"]]), int(counts[CNT["SAMPLES_VALID"]])
    fused = getattr(rnd, "dense", "f32") == "f16x2" and opt.K == 8
    ms_ch, ms_nb = stage_ms.get("chain", 0.0), stage_ms.get("mlp_neighbour", 0.0)
    # 2 x 256 x (284 + 256 + 263 + 256) + 2 x 256 flop per valid neighbour
    alg8d = 542720.0 * n_rows
    if fused and ms_ch > 0:
        n_small, n_tiny = int(counts[CNT["SAMPLES_SMALL"]]), int(counts[CNT["SAMPLES_TINY"]])
        rows_pad = 128 * ((n_valid - n_small - n_tiny + 15) // 16 + (n_small + 31) // 32 + (n_tiny + 63) // 64)
        issued = 3.0 * 2.0 * rows_pad * 256 * (64 + 256 + 272 + 256)
        # layers as executed (224 of block1.0's 284 columns live in the per-point table) + the alpha branch
        executed = 2.0 * n_rows * 256 * (60 + 256 + 263 + 256) + 2.0 * n_rows * 256
        sec = ms_ch * 1e-3
        ach_alg, ach_iss = alg8d / sec / 1e12, issued / sec / 1e12
        t_ch = [v for k, v in pmc.items() if "chain_ws_kernel" in k]
        busy, busy_src = chain_mfma_busy()
        return dict(
            kernel="chain_ws_kernel<0>: block1 -> block3 -> alpha + K-sums fused, weight-stationary (1 launch, %d "
                   "valid "
                   "neighbour rows in %d padded rows)" % (n_rows, rows_pad),
            bound="mfma", achieved=round(ach_alg, 1), peak=BF16_MFMA_PEAK_TF, unit="TFLOP/s",
            frac=round(ach_alg / BF16_MFMA_PEAK_TF, 4), achieved_issued=round(ach_iss, 1),
            frac_issued=round(ach_iss / BF16_MFMA_PEAK_TF, 4),
            frac_executed_fp32=round(executed / sec / 1e12 / BF16_MFMA_PEAK_TF, 4),
            traffic=int(t_ch[0]["hbm_bytes"]) if t_ch else None,
            traffic_source=("profiles/%s (%s)" % (TRAFFIC_JSON, PMC_NOTE)) if t_ch else None,
            mfma_busy=busy, mfma_busy_source=busy_src,
            flops_per_launch=alg8d, flops_issued_per_launch=issued, executed_fp32_flops_per_launch=executed,
            avg_launch_ms=round(ms_ch, 4), algorithmic_bytes_per_launch=int(n_rows * 168 + n_valid * 1028),
            row_padding_pct=round(100.0 * (rows_pad / max(n_rows, 1) - 1.0), 2),
            note="achieved / frac: ALGORITHMIC flops (SURVEY 8d, block1.0 counted with all 284 input columns) / "
                 "HIP-event time "
                 "/ the 2.5 PFLOP/s dense 16-bit peak.  achieved_issued: 16-bit MFMA flops issued (3 per fp32 product: "
                 "wm*xh + wh*xm + wh*xh, two-term fp16 split under exact power-of-two scales, fp32 accumulate; K and "
                 "row-slot "
                 "padding) -- what mfma_busy corroborates.  algorithmic bytes: 168 B per valid neighbour + 1028 B of "
                 "sums per "
                 "valid sample.",
            neighbour_stage=dict(chain_ms=round(ms_ch, 3), gather_ms=round(stage_ms.get("chain_gather", 0.0), 3)))
    if ms_nb > 0:
        flops_nb = 2.0 * n_rows * 256 * (284 + 256 + 263 + 256)
        flops_exec = 2.0 * n_rows * 256 * (60 + 256 + 263 + 256) if rnd.split_block1 else flops_nb
        ach = flops_nb / (ms_nb * 1e-3) / 1e12
        lin = {k: v for k, v in pmc.items() if "linear_f32_kernel<2, 2, 1, 0, 4" in k}
        n = sum(v["launches"] for v in lin.values())
        t_lin = sum(v["hbm_bytes"] * v["launches"] for v in lin.values()) / max(n, 1) if lin else None
        return dict(kernel="linear_f32_kernel<2,2,1,0,4,*> (block1+block3, 4 launches, M=%d rows)" % n_rows,
                    bound="mfma",
                    achieved=round(ach, 2), peak=F32_MFMA_PEAK_TF, unit="TFLOP/s", frac=round(ach / F32_MFMA_PEAK_TF,
                            4),
                    traffic=int(t_lin) if t_lin else None,
                    traffic_source=("profiles/%s (%s)" % (TRAFFIC_JSON, PMC_NOTE)) if t_lin else None,
                    flops_per_launch=flops_nb / 4, avg_launch_ms=round(ms_nb / 4, 4),
                    executed_tflops=round(flops_exec / (ms_nb * 1e-3) / 1e12, 2),
                    note="achieved = algorithmic flops / time; executed_tflops = MFMA flops actually issued / time")
    return None


def query_roofline(args, rnd, opt, cloud, sc, cam, dev, counts, stage_ms, CNT):
    """Point-query stage.  The in-frame time shares the GPU with the feature-pyramid rebuild on a side stream, so the
    roofline is quoted on the stage ALONE: HIP events around 5 back-to-back hnr_march_query launches on the bench frame,
    same buffers."""
    from hybridneuralrendering_amd import querier as Qm
    ms_q = stage_ms.get("query", 0.0)
    if ms_q <= 0:
        return None
    R, D, K = cam["raydir"].shape[0], opt.z_depth_dim, opt.K
    s_all, cells, cand = (int(counts[CNT[k]]) for k in ("SAMPLES", "CELLS_VISITED", "CANDIDATES"))
    # B_q = R (12 + ceil(D / 8) + 1) + s (12 + 27 * 4 + 4 K) + 4 cells + 16 candidates
    alg = R * (12 + (D + 7) // 8 + 1) + s_all * (12 + 27 * 4 + 4 * K) + 4 * cells + 16 * cand
    order = 1 if rnd.knn_order == "sorted" else 0
    knn = "knn_nb_kernel<8, %d" % order
    pmc = pmc_traffic() if profiled_workload(args) else {}
    t_q = [v for k, v in pmc.items() if "march_kernel" in k or knn in k]
    gbs = lambda ms: alg / (ms * 1e-3) / 1e9
    roof = dict(kernel="hnr_march_query: march_kernel + work-list scans + knn_nb_kernel<8,%d,2> (k-NN over the grid's "
                       "3x3x3 "
                       "neighbourhood lists)" % order,
                bound="hbm", peak=HBM_PEAK_GBS, unit="GB/s", achieved=round(gbs(ms_q), 1),
                frac=round(gbs(ms_q) / HBM_PEAK_GBS, 4),
                traffic=int(sum(v["hbm_bytes"] for v in t_q)) if len(t_q) == 2 else None,
                traffic_source=("profiles/%s (march_kernel + knn kernel, bytes per launch)" % TRAFFIC_JSON)
                if len(t_q) == 2 else None,
                algorithmic_bytes=int(alg), avg_launch_ms=round(ms_q, 4), in_frame_ms=round(ms_q, 4),
                neighbour_order=rnd.knn_order,
                per_ray=dict(samples=round(s_all / R, 2), cells_per_sample=round(cells / max(s_all, 1), 2),
                             candidates_per_sample=round(cand / max(s_all, 1), 2)))
    if opt.K != 8:
        return roof
    grid, hp = rnd.querier._grid_for(cloud.xyz[None])
    tmid = rnd.querier._tmid_for(float(sc.near), float(sc.far), D, R, dev)
    r2 = np.float32(hp[0] ** 2)
    ms_o = {}
    for o in (0, 1):
        q = lambda: Qm.march_query(grid, cam["campos"], cam["raydir"], tmid, opt.SR, K, r2, opt.kernel_size, pad=False,
                                   knn_order=o)
        q(); q()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            q()
        e1.record()
        torch.cuda.synchronize()
        ms_o[o] = e0.elapsed_time(e1) / 5
    roof.update(achieved=round(gbs(ms_o[order]), 1), frac=round(gbs(ms_o[order]) / HBM_PEAK_GBS, 4),
                avg_launch_ms=round(ms_o[order], 4),
                timing="HIP events around 5 back-to-back launches of the stage alone (in_frame_ms: beside the "
                       "feature-pyramid rebuild)")
    other = 1 - order
    roof["other_neighbour_order"] = dict(order="sorted" if other else "reference", avg_launch_ms=round(ms_o[other], 4),
                                         frac=round(gbs(ms_o[other]) / HBM_PEAK_GBS, 4))
    return roof


def amortised(rnd, agg, cloud, opt):
    """One-off work amortised over frames (rebuilt only when the cloud / the weights change), timed once."""
    from hybridneuralrendering_amd import querier as Q

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        return round(e0.elapsed_time(e1), 3)

    def records():
        rnd._rec_key = None
        rnd.point_records(cloud)
    hp = rnd.querier._hp
    build = lambda: Q.VoxelGrid(cloud.xyz, hp[2][:3], hp[5], hp[6], opt.query_size, opt.P, opt.max_o)
    # (the build hipMallocs ~1.2 GB: the first one of a process can take tens of ms on some boxes -- the better of two)
    return dict(grid_build_ms=min(timed(build), timed(build)), point_table_ms=timed(lambda: agg.point_table(cloud.emb)),
                point_records_ms=timed(records))


def resident_bytes(rnd, agg, cloud, cam, opt):
    """What one rank keeps in HBM for the frame (every rank of an N-GPU job replicates all of it): the cloud, the grid
    with its neighbourhood lists, the per-point table and records, the weight images, the reference-view feature map and
    the frame's workspace (sized for R * SR valid samples)."""
    import ctypes
    from hybridneuralrendering_amd import _lib
    nb = lambda t: int(t.numel() * t.element_size()) if isinstance(t, torch.Tensor) else 0
    R = cam["raydir"].shape[0]
    prm = _lib.RenderParams()
    prm.R, prm.SR, prm.K, prm.D, prm.V = R, int(opt.SR), int(opt.K), int(opt.z_depth_dim), 4
    prm.cap_samples = R * int(opt.SR)
    out = dict(cloud=sum(nb(t) for t in (cloud.xyz, cloud.emb, cloud.conf, cloud.dir, cloud.color)),
               grid=int((rnd.querier.last_grid_stats or {}).get("bytes", 0)),
               point_table=nb(rnd.point_table(cloud)), point_records=nb(rnd.point_records(cloud)),
               weight_images=nb(agg.packed_chain()) + sum(nb(m.packed) for m in agg.packed_mlp3().values() if hasattr(m,
                       "packed")),
               feature_map=nb(getattr(rnd, "_fm", None)),
               frame_workspace=int(_lib.lib().hnr_render_workspace_bytes(ctypes.byref(prm))))
    out["total"] = sum(out.values())
    return out


def fp32_anchor(args, rnd, opt, agg, cloud, cam, sc, dev, col):
    """Same-run anchor for `dtype: f32`: the identical frame with every per-neighbour layer on fp32 MFMA (HNR_DENSE=f32:
    v_mfma_f32_32x32x2_f32, per-stage calls), one warm-up + one timed frame, and its largest colour difference from the
    frame of the timed region."""
    from hybridneuralrendering_amd.render import HybridRenderer
    old_env = os.environ.get("HNR_DENSE")
    os.environ["HNR_DENSE"] = "f32"
    try:
        rnd32 = HybridRenderer(opt, agg, dev)
        rnd32.knn_order = rnd.knn_order
        render_frame(rnd32, cloud, cam, sc, args.chunk)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        col32, _ = render_frame(rnd32, cloud, cam, sc, args.chunk)
        torch.cuda.synchronize()
        return dict(fp32_mfma_ms_per_step=round((time.perf_counter() - ta) * 1e3, 3),
                    max_abs_vs_f16x2_frame=float((col32 - col).abs().max()),
                    note="HNR_DENSE=f32: the per-neighbour layers as four fp32-MFMA launches; same frame, same process")
    finally:
        if old_env is None:
            os.environ.pop("HNR_DENSE", None)
        else:
            os.environ["HNR_DENSE"] = old_env


DENSE_NOTE = {"f32": "fp32 MFMA (v_mfma_f32_32x32x2_f32)",
              "f16x2": "per-neighbour chain fused in one kernel: fp32 operands split into 2 fp16 terms under exact "
                       "power-of-two row "
                       "/ layer scales, 3 fp16 MFMAs per product, fp32 accumulate (error vs fp64 at or below the "
                       "fp32-MFMA path's, "
                       "tests/test_chain_gpu.py); all other layers fp32 MFMA"}


def describe_config(args, sc, opt, rnd, R_frame, R_job, R, world, strong):
    scene = {"scene0241": "scene0241_01-like room", "scene0101": "scene0101_04-like room"}.get(args.scene,
            args.scene + "-like object")
    fused = getattr(rnd, "dense", "f32") == "f16x2" and opt.K == 8
    how = "scan lines dealt round-robin" if args.shard == "lines" else "contiguous scan-line blocks"
    return {
        "workload": "%s synthetic scene (SURVEY 8d): %d points, %dx%d frame margin %d = %d rays per step (%s), SR=%d "
                    "K=%d P=%d "
                    "max_o=%d D=%d, 4 reference views %dx%d, hybrid viewmlp forward "
                    "(query+gather+aggregate+composite), neighbour "
                    "lists in %s order; random-init weights with alpha_branch.0 rescaled (weight x30, bias = 30) so "
                    "that opacities "
                    "spread over (0,1)" % (scene, sc.xyz.shape[0], sc.w, sc.h, args.margin, R_frame,
                                           "ONE fixed frame sharded over the ranks" if strong else "one such frame per "
                                                                                                   "rank",
                                           opt.SR, opt.K, opt.P, opt.max_o, opt.z_depth_dim, sc.h, sc.w, rnd.knn_order),
        "entry": ("hnr_render_forward (one library call per frame, no host read)" if getattr(rnd, "single_call",
                False) and fused
                  else "per-stage C-ABI calls from Python"),
        "rays_per_step": R_job, "rays_per_gpu": R, "points": int(sc.xyz.shape[0]),
        "chunk_rays": args.chunk if args.chunk > 0 else R,
        "parallelism": ("one fixed frame ray-sharded x%d (%s), one RCCL gather" % (world, how)) if strong
                       else "one frame per rank x%d, one RCCL gather" % world}
