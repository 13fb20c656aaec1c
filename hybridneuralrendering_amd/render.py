"""The fused forward render path: query -> gather/aggregate -> composite, all in libhnr_hip.so.

`HybridRenderer.render_rays` is the MI355X counterpart of one pass of
`NeuralPointsRayMarching.forward` + `fill_invalid`
(/root/reference/models/neural_points_volumetric_model.py:257-391, :87-126) over R rays, without the
reference's per-chunk grid rebuild, raypos tensor, boolean-mask copies or host syncs (one host read of
three counters per launch sizes the MLP workspaces).
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import HnrError, CNT
from . import querier as Q


def _i32(n, dev):
    return torch.empty((max(int(n), 1),), dtype=torch.int32, device=dev)


def _f32(shape, dev):
    return torch.empty(shape, dtype=torch.float32, device=dev)


class _Stage:
    """Optional HIP-event bracket around a stage (events are recorded on the launch stream)."""

    def __init__(self, timers, name):
        self.timers, self.name = timers, name

    def __enter__(self):
        if self.timers is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.timers is not None:
            self.e1.record()
            self.timers.setdefault(self.name, []).append((self.e0, self.e1))
        return False


class PointCloud:
    """The neural point buffers in the kernels' layouts (views of the NeuralPoints parameters)."""

    def __init__(self, xyz, emb, conf, pdir, color):
        g = _lib.require_gpu
        self.xyz = g(xyz.detach(), "xyz", torch.float32).reshape(-1, 3)
        n = self.xyz.shape[0]
        self.emb = g(emb.detach(), "points_embeding", torch.float32).reshape(n, -1)
        self.conf = g(conf.detach(), "points_conf", torch.float32).reshape(n)
        self.dir = g(pdir.detach(), "points_dir", torch.float32).reshape(n, 3)
        self.color = g(color.detach(), "points_color", torch.float32).reshape(n, 3)
        self.F = self.emb.shape[1]


class HybridRenderer:
    def __init__(self, opt, aggregator, device):
        self.opt = opt
        self.agg = aggregator
        self.device = torch.device(device)
        self.querier = Q.lighting_fast_querier(self.device, opt)
        self._fm_key, self._fm, self._fm_src = None, None, None
        self._pt_key, self._pt, self._pt_src = None, None, None
        self.split_block1 = True          # fold the point-only 224 columns of block1.0 into a per-point table
        # dense arithmetic of the per-neighbour layers (fp32 in / out, fp32-class error in every mode):
        #   "f16x2"  (default) the whole per-neighbour chain in ONE kernel (hnr_chain_forward, K = 8): operands split into two fp16
        #            terms, three 16-bit MFMAs per product, activations never leave the CU;
        #   "f32"    one launch per layer on fp32 MFMA (hnr_linear_f32).
        self.dense = os.environ.get("HNR_DENSE", "f16x2")
        if self.dense not in ("f16x2", "f32"):
            raise HnrError("HNR_DENSE must be f16x2 or f32, got %r" % self.dense)
        self.split_merge = True           # multiply the colour-feature columns of aux_merge_weight_block.0 once per sample
        # the whole frame as ONE library call (hnr_render_forward: no host read between query and composite); HNR_SINGLE_CALL=0 runs the
        # same kernels stage by stage from Python with exactly sized buffers (one host read of the counters)
        self.single_call = os.environ.get("HNR_SINGLE_CALL", "1") != "0"
        # neighbour order inside a sample's K slots: "reference" = exactly the reference kernel's (insertion history), "sorted" = the same
        # neighbour SET in ascending-distance order (hnr_query_params.knn_order = 1: cheaper insertion; colours agree to fp32 summation order)
        self.knn_order = os.environ.get("HNR_KNN_ORDER", "reference")
        if self.knn_order not in ("reference", "sorted"):
            raise HnrError("HNR_KNN_ORDER must be 'reference' or 'sorted' (got %r)" % (self.knn_order,))
        self.fuse_mixup = True            # hnr_mixup_stage instead of the mix-up MLP + hnr_final_color (staged path switch)
        self.fuse_merge = True            # V = 4: hnr_merge_stage instead of hnr_proj_rows + the merge-weight MLP + hnr_merge (staged path switch)
        self.last_counts = None
        if getattr(opt, "which_render_func", "radiance") != "radiance" or getattr(opt, "which_blend_func", "alpha") != "alpha" \
                or getattr(opt, "which_tonemap_func", "off") != "off":
            raise HnrError("only radiance render / alpha blend / no tone map are implemented (all shipped configs)")

    # -- per-frame reference-view features, cached -------------------------------------------------
    def feature_map(self, images_nearest):
        ev = getattr(self, "_fm_event", None)
        if ev is not None:
            torch.cuda.current_stream(images_nearest.device).wait_event(ev)      # (a map built on the side stream: order this stream behind it)
        key = (images_nearest.data_ptr(), tuple(images_nearest.shape), images_nearest._version,
               tuple((p.data_ptr(), p._version) for p in self.agg.parameters()))
        if key != self._fm_key:
            self._fm = self.agg.image_features(images_nearest)
            self._fm_key = key
            self._fm_event = None                   # built on the calling stream (feature_map_async records an event when it builds on the side stream)
            # keep the keyed tensor alive: a freed buffer's address (and version 0) can be handed to the NEXT frame's images by the
            # caching allocator, which would look like a cache hit
            self._fm_src = images_nearest
        return self._fm

    def feature_map_async(self, images_nearest):
        """(feature map, event or None): a cached map needs no event; a rebuild is issued on a side stream (after everything already queued on the
        current stream, so its inputs are complete) and the returned event marks its end."""
        key = (images_nearest.data_ptr(), tuple(images_nearest.shape), images_nearest._version,
               tuple((p.data_ptr(), p._version) for p in self.agg.parameters()))
        if key == self._fm_key:
            # a cached map may still be in flight on the side stream (built for a launch on another stream, or for a launch that fell back to the
            # staged path): hand its event out until the consumer has been ordered behind it -- an extra wait on a complete event costs nothing
            return self._fm, getattr(self, "_fm_event", None)
        cur = torch.cuda.current_stream(images_nearest.device)
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = torch.cuda.Stream(device=images_nearest.device)
        side = self._side_stream
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            fm = self.feature_map(images_nearest)
            ev = torch.cuda.Event()
            ev.record(side)
        fm.record_stream(cur)                       # allocated under the side stream, consumed on the launch stream
        self._fm_event = ev
        return fm, ev

    def point_table(self, cloud):
        """Per-point addend of block1's first layer, rebuilt when the embeddings or the weights change.  The table lives in storage with 25 % slack:
        a cloud that only GREW (NeuralPoints.grow_points appends; the old rows are verified unchanged on the device) gets the rows of its new points
        computed behind the old ones instead of all N rows again (3 ms at 2 M points)."""
        wkey = tuple((p.data_ptr(), p._version) for p in self.agg.block1.parameters())
        key = (cloud.emb.data_ptr(), tuple(cloud.emb.shape), cloud.emb._version, wkey)
        if key == self._pt_key:
            return self._pt
        n, store = int(cloud.emb.shape[0]), getattr(self, "_pt_store", None)
        old = self._pt_src
        grown = (store is not None and self._pt_key is not None and self._pt_key[3] == wkey and old is not None and old.shape[0] < n <= store.shape[0]
                 and old.shape[1:] == cloud.emb.shape[1:] and bool(torch.equal(cloud.emb[:old.shape[0]], old)))      # (one host read, on the grow path only)
        if grown:
            n_old = int(old.shape[0])
            self.agg.point_table(cloud.emb[n_old:].contiguous(), out=store[n_old:n])
        else:
            store = torch.empty((n + n // 4 + 1024, 256), dtype=torch.float32, device=cloud.emb.device)
            self.agg.point_table(cloud.emb, out=store[:n])
            self._pt_store = store
        self._pt, self._pt_key = store[:n], key
        self._pt_src = cloud.emb                                 # same reason as in feature_map
        return self._pt

    def point_records(self, cloud):
        """hnr_point_records image of the cloud's xyz / conf / dir / colour buffers (48 B per point: what the fused gather reads per
        neighbour), rebuilt when any of them changes."""
        src = (cloud.xyz, cloud.conf, cloud.dir, cloud.color)
        key = tuple((t.data_ptr(), tuple(t.shape), t._version) for t in src)
        if key != getattr(self, "_rec_key", None):
            n = int(cloud.xyz.reshape(-1, 3).shape[0])
            rec = torch.empty((n, 12), dtype=torch.float32, device=cloud.xyz.device)
            with torch.cuda.device(cloud.xyz.device):
                _lib.check(_lib.lib().hnr_point_records(_lib.ptr(cloud.xyz), _lib.ptr(cloud.conf), _lib.ptr(cloud.dir), _lib.ptr(cloud.color), n,
                                                        _lib.ptr(rec), _lib.stream()), "hnr_point_records")
            self._rec, self._rec_key, self._rec_src = rec, key, src      # the sources stay referenced: a recycled address is not a stale hit
        return self._rec

    # -- stage 3: gather + aggregate ----------------------------------------------------------------
    def aggregate(self, cloud, qres, raydir, campos, camrot, w2c_nearest, intrinsic_nearest, campos_nearest, featmap,
                  frame_weight=None, want_weights=False, timers=None):
        """Returns decoded [R,SR,4] (sigma, r, g, b; zeros where the sample has no neighbour)."""
        L = _lib.lib()
        pk = self.agg.packed()
        dev = raydir.device
        pidx, loc_w, counts, work = qres["sample_pidx"], qres["sample_loc_w"], qres["counts"], qres["work"]
        R, SR, K = pidx.shape
        decoded = torch.zeros((R, SR, 4), dtype=torch.float32, device=dev)
        out = dict(decoded=decoded)
        if R == 0:
            return out
        st = _lib.stream
        # the one host read: sizes of the packed row buffers
        c = counts.cpu()
        self.last_counts = c
        n_valid, n_rows = int(c[CNT["SAMPLES_VALID"]]), int(c[CNT["NEIGHBOURS"]])
        if n_valid == 0:
            return out
        vs_item, vs_off, vs_cnt = _i32(n_valid, dev), _i32(n_valid, dev), _i32(n_valid, dev)
        scratch = _i32(3 * ((R * SR + 1023) // 1024) + 3, dev)
        overflow = torch.zeros(1, dtype=torch.int32, device=dev)
        p = _lib.ptr
        T = lambda name: _Stage(timers, name)
        with torch.cuda.device(dev):
          fused = self.dense == "f16x2" and K == 8
          with T("plan_gather"):
            if fused:
                # valid samples with more than four neighbours first, then the small ones (4 row slots each in the chain's row tiles)
                _lib.check(L.hnr_chain_plan(p(work), p(pidx), p(counts), K, R * SR, int(L.hnr_chain_classes()), p(vs_item), n_valid,
                                            p(scratch), st()), "hnr_chain_plan")
            else:
                _lib.check(L.hnr_sample_plan(p(work), p(pidx), p(counts), K, R * SR, p(vs_item), p(vs_off), p(vs_cnt), n_valid, n_rows,
                                             p(scratch), p(overflow), st()), "hnr_sample_plan")
          # memory guard (the per-layer path keeps 3360 B per neighbour row, the fused chain 296 B per neighbour SLOT + ~3 KB per sample):
          # a frame that cannot fit is refused with the remedy named instead of dying in the allocator
          need = (n_valid * 8 * 296 + n_valid * 6000) if fused else (n_rows * 3400 + n_valid * 6000)
          free, _tot = torch.cuda.mem_get_info(dev)
          cached = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
          if need > 0.9 * (free + cached):
              raise HnrError("render_rays: this launch needs ~%.1f GB of workspace for %d valid samples / %d neighbour rows but only %.1f GB are "
                             "free; render the frame in chunks (driver.render_image(chunk_rays=...), bench.py --chunk)" % (need / 1e9, n_valid, n_rows, (free + cached) / 1e9))
          if fused:
            # the fused chain: gather + geometry + PE -> operand image, then block1 -> block3 -> alpha + K-sums in one kernel
            ws = torch.empty((int(L.hnr_chain_workspace_bytes(n_valid)),), dtype=torch.uint8, device=dev)
            X5 = _f32((n_valid, 280), dev)
            sigma = _f32((n_valid,), dev)
            w_out = c_out = None
            if want_weights:
                w_out = torch.zeros((R, SR, K), dtype=torch.float32, device=dev)
                c_out = cloud.conf[0].clamp(0.0001, 1.0).expand(R, SR, K).contiguous()
            ptab = self.point_table(cloud)
            with T("chain_gather"):
                _lib.check(L.hnr_chain_gather_rec(p(self.point_records(cloud)), p(pidx), p(loc_w), p(raydir),
                                                  p(campos), p(camrot), p(vs_item), p(counts), SR, K, n_valid, p(ws), p(X5), 280,
                                                  p(w_out) if want_weights else None, p(c_out) if want_weights else None, st()),
                           "hnr_chain_gather_rec")
            with T("chain"):
                _lib.check(L.hnr_chain_forward(p(ws), p(ptab), int(ptab.stride(0)), p(self.agg.packed_chain()), p(counts), n_valid,
                                               float(pk["slope"]), p(X5), 280, p(sigma), None, 0, st()), "hnr_chain_forward")
            del ws
          else:
            split = self.split_block1
            A = _f32((n_rows, 256 if split else 284), dev)       # (X1,) later H3
            B = _f32((n_rows, 256), dev)       # H1, later H4
            C = _f32((n_rows, 264), dev)       # X3 = [H2 | extras]
            wagg = _f32((n_rows,), dev)
            Xd = _f32((n_rows, 64), dev) if split else None      # PE5(dists6), 60 columns used
            row_pid = _i32(n_rows, dev) if split else None
            ptab = self.point_table(cloud) if split else None
            w_out = c_out = None
            if want_weights:
                w_out = torch.zeros((R, SR, K), dtype=torch.float32, device=dev)
                # empty slots read point 0 in the reference (index clamp, neural_points.py:711): same value here
                c_out = cloud.conf[0].clamp(0.0001, 1.0).expand(R, SR, K).contiguous()
            _lib.check(L.hnr_gather_rows(p(cloud.xyz), p(cloud.emb), p(cloud.conf), p(cloud.dir), p(cloud.color), cloud.F,
                                         p(pidx), p(loc_w), p(raydir), p(campos), p(camrot), p(vs_item), p(vs_off), p(vs_cnt),
                                         p(counts), SR, K, n_valid, p(Xd) if split else p(A), 64 if split else 284, p(C), 264, p(wagg),
                                         p(w_out) if want_weights else None, p(c_out) if want_weights else None,
                                         p(row_pid) if split else None, st()),
                       "hnr_gather_rows")
            sl = pk["slope"]
            with T("mlp_neighbour"):
              if split:
                  pk["b1_dist"].gather_add(Xd, ptab, row_pid, out=B, act=True, slope=sl, K=60)   # 60 -> 256 (+ per-point addend)
              else:
                  pk["b1"][0](A, out=B, act=True, slope=sl)                   # 284 -> 256
              l12, l30, l32 = pk["b1"][1], pk["b3"][0], pk["b3"][1]
              with T("dense_b1_2"):
                  l12(B, out=C, act=True, slope=sl)                           # 256 -> 256 into X3[:, :256]
              H3 = A[:, :256]
              with T("dense_b3_0"):
                  l30(C, out=H3, act=True, slope=sl, K=263)                   # 263 -> 256
              with T("dense_b3_2"):
                  l32(H3, out=B, act=True, slope=sl)                          # 256 -> 256  (H4)
            with T("ksum"):
              X5 = _f32((n_valid, 280), dev)
              sigma = _f32((n_valid,), dev)
              _lib.check(L.hnr_ksum(p(B), 256, p(wagg), p(pk["alpha_w"]), p(pk["alpha_b"]), p(vs_item), p(vs_off), p(vs_cnt),
                                    p(raydir), p(counts), SR, n_valid, p(X5), 280, p(sigma), st()), "hnr_ksum")
            del A, B, C, Xd
          sl = pk["slope"]
          fused_s = self.dense == "f16x2"          # the per-sample MLPs as fused three-layer launches (hnr_mlp3_forward)
          m3 = self.agg.packed_mlp3() if fused_s else None
          ci = CNT["SAMPLES_VALID"]
          with T("mlp_colorfeat"):
            if fused_s:
                pre = _f32((n_valid, 64), dev)            # colour-feature part of aux_merge_weight_block.0 (bias included), once per sample
                CF = m3["cf"](X5, _f32((n_valid, 128), dev), n_valid, counts, ci, 1, slope=sl, out2=pre)
            else:
                T1, T2 = _f32((n_valid, 128), dev), _f32((n_valid, 128), dev)
                pk["cf"][0](X5, out=T1, act=True, slope=sl)
                pk["cf"][1](T1, out=T2, act=True, slope=sl)
                CF = pk["cf"][2](T2, out=T1, act=True, slope=sl)
          no_views = getattr(self.opt, "use_nearest", 4) == 0
          if no_views:
            # use_nearest = 0 (scene241.sh): the image branch is off, merged = 0 (point_aggregators.py:1257-1258)
            X7 = torch.zeros((n_valid, 92), dtype=torch.float32, device=dev)
            X7[:, :45] = CF[:, :45]
          else:
            V, H, W = featmap.shape[0], featmap.shape[1], featmap.shape[2]
            fw = None if frame_weight is None else _lib.require_gpu(frame_weight, "frame_weight", torch.float32).reshape(-1)
            if fused_s and self.split_merge and V == 4 and self.fuse_merge:
              # reprojection + feature gather + merge-weight MLP + weighted merge in one launch (hnr_merge_stage)
              with T("mlp_merge"):
                X7 = _f32((n_valid, 92), dev)
                _lib.check(L.hnr_merge_stage(p(loc_w), p(vs_item), p(counts), p(w2c_nearest), p(intrinsic_nearest), p(campos), p(campos_nearest),
                                             p(featmap), V, H, W, p(fw) if fw is not None else None, p(pre), 64, p(m3["mw"].packed), p(pk["mw_last_w"]),
                                             p(pk["mw_last_b"]), p(CF), 128, n_valid, float(sl), p(X7), 92, st()), "hnr_merge_stage")
            else:
              with T("proj_rows"):
                ld6 = 48 if self.split_merge else 176
                X6 = _f32((V * n_valid, ld6), dev)
                vmask = _f32((V * n_valid,), dev)
                row_s = _i32(V * n_valid, dev) if self.split_merge else None
                _lib.check(L.hnr_proj_rows(p(loc_w), p(vs_item), p(counts), p(w2c_nearest), p(intrinsic_nearest), p(campos),
                                           p(campos_nearest), p(featmap), V, H, W, p(CF), 128, n_valid, p(X6), ld6, p(vmask),
                                           p(row_s) if self.split_merge else None, st()), "hnr_proj_rows")
              with T("mlp_merge"):
                M1 = _f32((V * n_valid, 64), dev)
                if fused_s and self.split_merge:
                    m3["mw"](X6, M1, V * n_valid, counts, ci, V, slope=sl, R=pre, ridx=row_s)
                elif self.split_merge:
                    M2 = _f32((V * n_valid, 64), dev)
                    pre = pk["mw0_cf"](CF, act=False)                                     # [S,64] once per sample (bias included)
                    pk["mw0_fd"].gather_add(X6, pre, row_s, out=M1, act=True, slope=sl)   # 48 -> 64 per (view, sample) + addend
                else:
                    M2 = _f32((V * n_valid, 64), dev)
                    pk["mw"][0](X6, out=M1, act=True, slope=sl)
                if not (fused_s and self.split_merge):
                    pk["mw"][1](M1, out=M2, act=True, slope=sl)
                    pk["mw"][2](M2, out=M1, act=True, slope=sl)
              with T("merge"):
                X7 = _f32((n_valid, 92), dev)
                _lib.check(L.hnr_merge(p(X6), ld6, p(M1), 64, p(pk["mw_last_w"]), p(pk["mw_last_b"]), p(vmask),
                                       p(fw) if fw is not None else None, p(CF), 128, p(counts), V, n_valid, p(X7), 92,
                                       None, None, 0, st()), "hnr_merge")
          if fused_s and self.fuse_mixup:
            # color_mixup_block + residual + color_final_block + decode in one launch (hnr_mixup_stage)
            with T("mlp_mixup"):
                _lib.check(L.hnr_mixup_stage(p(X7), 92, p(m3["mx"].packed), p(CF), 128, p(pk["fin_w"]), p(pk["fin_b"]), p(sigma), p(vs_item), p(counts),
                                             n_valid, float(sl), None, 0, p(decoded), st()), "hnr_mixup_stage")
          else:
           with T("mlp_mixup"):
            Y1 = _f32((n_valid, 48), dev)
            if fused_s:
                m3["mx"](X7, Y1, n_valid, counts, ci, 1, slope=sl)
            else:
                Y2 = _f32((n_valid, 48), dev)
                pk["mx"][0](X7, out=Y1, act=True, slope=sl, K=90)
                pk["mx"][1](Y1, out=Y2, act=True, slope=sl, K=45)
                pk["mx"][2](Y2, out=Y1, act=False, K=45)
           with T("final_color"):
            _lib.check(L.hnr_final_color(p(Y1), 48, p(CF), 128, p(pk["fin_w"]), p(pk["fin_b"]), p(sigma), p(vs_item), p(counts),
                                         n_valid, p(decoded), st()), "hnr_final_color")
        out["overflow"] = overflow                 # checked by render_rays AFTER the composite is queued (no pipeline drain here)
        if want_weights:
            out.update(weight=w_out, conf_coefficient=c_out)
        return out

    # -- stage 4 ------------------------------------------------------------------------------------
    def composite(self, decoded, qres, campos, camrot, bg_color, want_blend=False):
        L = _lib.lib()
        pidx, loc_w, mask = qres["sample_pidx"], qres["sample_loc_w"], qres["ray_mask"]
        R, SR, K = pidx.shape
        dev = pidx.device
        col, opa, isbg = _f32((R, 3), dev), _f32((R, SR), dev), _f32((R,), dev)
        bw = _f32((R, SR), dev) if want_blend else None
        p = _lib.ptr
        with torch.cuda.device(dev):
            _lib.check(L.hnr_composite(p(decoded), p(loc_w), p(pidx), p(mask), None if qres.get("padded", True) else p(qres["ray_nsamp"]),
                                       p(campos), p(camrot), p(bg_color), R, SR, K,
                                       float(np.float32(self.opt.vsize[2])), int(getattr(self.opt, "raydist_mode_unit", 0) > 0),
                                       p(col), p(opa), p(isbg), p(bw) if want_blend else None, _lib.stream()), "hnr_composite")
        return dict(coarse_raycolor=col, coarse_point_opacity=opa, coarse_is_background=isbg, blend_weight=bw)

    # -- the whole path as ONE library call (no host read between query and composite) ---------------
    def _single_call(self, cloud, raydir, campos, camrot, bg_color, tmid, grid, radius2, w2c_nearest, campos_nearest, intrinsic_nearest, fm,
                     frame_weight, want_weights, timers, fm_ready=None):
        """hnr_render_forward: every launch of the frame issued by the library on the current stream; workspaces sized for the worst
        case R*SR valid samples (or what fits: the status word reports an overflow), no `.cpu()` / `.item()` on the way."""
        L, p = _lib.lib(), _lib.ptr
        opt, dev = self.opt, raydir.device
        R, SR, K = raydir.shape[0], int(opt.SR), int(opt.K)
        V = 0 if fm is None else int(fm.shape[0])
        prm = _lib.RenderParams()
        prm.R, prm.SR, prm.K, prm.D = R, SR, K, int(tmid.shape[-1])
        prm.tmid_stride = 0 if tmid.dim() == 1 else int(tmid.shape[1])
        for i in range(3):
            prm.kernel_size[i] = int(opt.kernel_size[i])
        prm.radius2, prm.vsize_z = float(radius2), float(np.float32(opt.vsize[2]))
        prm.raydist_mode_unit, prm.V = int(getattr(opt, "raydist_mode_unit", 0) > 0), V
        cap = R * SR
        prm.cap_samples = cap
        prm.knn_order = 1 if self.knn_order == "sorted" else 0
        nbytes = int(L.hnr_render_workspace_bytes(ctypes.byref(prm)))
        free, _total = torch.cuda.mem_get_info(dev)
        cached = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
        if nbytes > 0.6 * (free + cached):
            return None                                   # the worst case does not fit: the staged path sizes its buffers exactly
        ws = torch.empty((nbytes + 256,), dtype=torch.uint8, device=dev)
        off = (-ws.data_ptr()) % 256
        pk, agg = self.agg.packed(), self.agg
        m3 = agg.packed_mlp3()
        ptab = self.point_table(cloud)
        cl = _lib.RenderCloud(p(cloud.xyz), p(cloud.conf), p(cloud.dir), p(cloud.color), p(ptab), int(ptab.stride(0)), p(self.point_records(cloud)))
        wt = _lib.RenderWeights(p(agg.packed_chain()), p(m3["cf"].packed), p(m3["mw"].packed), p(m3["mx"].packed),
                                p(pk["mw_last_w"]), p(pk["mw_last_b"]), p(pk["fin_w"]), p(pk["fin_b"]), float(pk["slope"]))
        cam = _lib.RenderCamera(p(campos), p(camrot), p(raydir), p(tmid), p(bg_color))
        vw = None
        fw = None if frame_weight is None else _lib.require_gpu(frame_weight, "frame_weight", torch.float32).reshape(-1)
        if V > 0:
            vw = _lib.RenderViews(p(w2c_nearest), p(intrinsic_nearest), p(campos_nearest), p(fm), int(fm.shape[1]), int(fm.shape[2]),
                                  p(fw) if fw is not None else None, ctypes.c_void_p(fm_ready.cuda_event) if fm_ready is not None else None)
        col, opa, isbg = _f32((R, 3), dev), _f32((R, SR), dev), _f32((R,), dev)
        bw = _f32((R, SR), dev) if want_weights else None
        mask = torch.empty((R,), dtype=torch.int8, device=dev)
        decoded = _f32((R, SR, 4), dev)
        pidx = torch.empty((R, SR, K), dtype=torch.int32, device=dev)
        loc = _f32((R, SR, 3), dev)
        nsamp = torch.empty((R,), dtype=torch.int32, device=dev)
        counts = torch.empty((_lib.NCOUNTS,), dtype=torch.int64, device=dev)
        status = torch.empty((2,), dtype=torch.int32, device=dev)
        w_out = c_out = None
        if want_weights:
            w_out = torch.zeros((R, SR, K), dtype=torch.float32, device=dev)
            c_out = cloud.conf[0].clamp(0.0001, 1.0).expand(R, SR, K).contiguous()     # empty slots read point 0 in the reference (:711)
        ev = None
        if timers is not None:
            ev = _lib.StageEvents()
            timers.setdefault("_stage_events", []).append(ev)
        out = _lib.RenderOutputs(p(col), p(opa), p(isbg), p(bw) if bw is not None else None, p(mask), p(decoded), p(pidx), p(loc), p(nsamp), p(counts),
                                 p(status), p(w_out) if w_out is not None else None, p(c_out) if c_out is not None else None,
                                 ev.arr if ev is not None else None)
        with torch.cuda.device(dev):
            _lib.check(L.hnr_render_forward(grid.handle, ctypes.byref(prm), ctypes.byref(cl), ctypes.byref(wt), ctypes.byref(cam),
                                            ctypes.byref(vw) if vw is not None else None, ctypes.c_void_p(ws.data_ptr() + off), nbytes,
                                            ctypes.byref(out), _lib.stream()), "hnr_render_forward")
        self._keepalive = (ws, fw)             # the launches are asynchronous: the workspace must outlive them (next call replaces it)
        res = dict(coarse_raycolor=col, coarse_point_opacity=opa, coarse_is_background=isbg, blend_weight=bw, ray_mask=mask, decoded=decoded,
                   sample_pidx=pidx, sample_loc_w=loc, ray_nsamp=nsamp, counts=counts, status=status)
        if want_weights:
            res.update(weight=w_out, conf_coefficient=c_out)
        return res

    @staticmethod
    def check_status(outs):
        """Reads the status word(s) of single-call launches (ONE host read for any number of launches): raises when a workspace capacity was
        exceeded -- the frame would be missing samples.  `outs`: an output dict of render_rays or a list of them (dicts of the staged path carry
        no status word: their buffers are sized exactly)."""
        if isinstance(outs, dict):
            outs = [outs]
        st = [o["status"] for o in outs if isinstance(o, dict) and o.get("status") is not None]
        if not st:
            return
        st = torch.stack(st).cpu()
        bad = (st[:, 0] != 0).nonzero().reshape(-1)
        if bad.numel():
            i = int(bad[0])
            raise HnrError("hnr_render_forward: launch %d of %d produced %d valid shading samples, more than its workspace capacity; the extra samples "
                           "were dropped (render in chunks: driver.render_image(chunk_rays=...))" % (i, st.shape[0], int(st[i, 1])))

    # -- the whole path -------------------------------------------------------------------------------
    def render_rays(self, cloud, raydir, campos, camrot, bg_color, near, far, c2w_nearest, campos_nearest, intrinsic_nearest,
                    images_nearest, frame_weight=None, want_weights=False, w2c_nearest=None, timers=None, pad=False):
        """raydir [R,3]; campos [3]; camrot [3,3]; c2w_nearest [V,4,4]; images_nearest [V,H,W,3] (or with a leading 1).
        Returns full-R outputs (fill_invalid applied): coarse_raycolor [R,3], coarse_point_opacity [R,SR],
        coarse_is_background [R], ray_mask [R] i8, decoded [R,SR,4] + the query tensors."""
        g = _lib.require_gpu
        raydir = g(raydir, "raydir", torch.float32).reshape(-1, 3)
        campos = g(campos, "campos", torch.float32).reshape(3)
        camrot = g(camrot, "camrotc2w", torch.float32).reshape(3, 3)
        bg_color = g(bg_color, "bg_color", torch.float32).reshape(3)
        c2w_nearest = g(c2w_nearest, "c2w_nearest", torch.float32).reshape(-1, 4, 4)
        campos_nearest = g(campos_nearest, "campos_nearest", torch.float32).reshape(-1, 3)
        intrinsic_nearest = g(intrinsic_nearest, "intrinsic_nearest", torch.float32).reshape(3, 3)
        if w2c_nearest is None:
            w2c_nearest = torch.inverse(c2w_nearest)          # 4x4 plumbing op (:250); V matrices per frame
        w2c_nearest = w2c_nearest.contiguous()
        q = self.querier
        grid, hp = q._grid_for(cloud.xyz[None])
        tmid = q._tmid_for(float(near), float(far), self.opt.z_depth_dim, raydir.shape[0], raydir.device)
        if self.dense == "f16x2" and self.opt.K == 8 and self.single_call and not pad and raydir.shape[0] > 0:
            # the reference-view feature pyramid does not depend on the rays: a rebuild (new reference views) runs on a side stream under the
            # query and the per-neighbour chain; the library makes the launch stream wait for it right before the merge stage
            fm, fm_ready = None, None
            if getattr(self.opt, "use_nearest", 4) != 0:
                with _Stage(timers, "featmap"):
                    fm, fm_ready = self.feature_map_async(images_nearest)
            res = self._single_call(cloud, raydir, campos, camrot, bg_color, tmid, grid, np.float32(hp[0] ** 2), w2c_nearest, campos_nearest,
                                    intrinsic_nearest, fm, frame_weight, want_weights, timers, fm_ready=fm_ready)
            if res is not None:
                return res
            if fm_ready is not None:
                # the single call did not run (workspace larger than the free memory): the staged path below reads the map on the current stream
                torch.cuda.current_stream(raydir.device).wait_event(fm_ready)
        with _Stage(timers, "query"):
            # pad=False: only kept slots are written (no -1 / 0 padding stores); everything downstream takes ray_nsamp
            qres = Q.march_query(grid, campos, raydir, tmid, self.opt.SR, self.opt.K, np.float32(hp[0] ** 2), self.opt.kernel_size, pad=pad,
                                 knn_order=1 if (self.knn_order == "sorted" and self.opt.K == 8) else 0)
        with _Stage(timers, "featmap"):
            fm = None if getattr(self.opt, "use_nearest", 4) == 0 else self.feature_map(images_nearest)
        a = self.aggregate(cloud, qres, raydir, campos, camrot, w2c_nearest, intrinsic_nearest, campos_nearest, fm,
                           frame_weight=frame_weight, want_weights=want_weights, timers=timers)
        with _Stage(timers, "composite"):
            out = self.composite(a["decoded"], qres, campos, camrot, bg_color, want_blend=want_weights)
        if "overflow" in a and int(a["overflow"].item()) != 0:
            raise HnrError("hnr_sample_plan: row buffers too small (internal sizing error)")
        out.update(ray_mask=qres["ray_mask"], decoded=a["decoded"], sample_pidx=qres["sample_pidx"],
                   sample_loc_w=qres["sample_loc_w"], ray_nsamp=qres["ray_nsamp"], counts=qres["counts"])
        if want_weights:
            out.update(weight=a.get("weight"), conf_coefficient=a.get("conf_coefficient"))
        return out
