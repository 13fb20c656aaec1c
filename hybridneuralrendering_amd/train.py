"""Training step of the fused path: forward with saved activations + hand-written HIP backward, wrapped in one
torch.autograd.Function so the reference's shell (losses, optimisers, schedulers) works unchanged.

Counterpart of NeuralPointsRayMarching.forward in train mode (/root/reference/models/neural_points_volumetric_model.py:257-427:
jittered depths models/neural_points/query_point_indices_worldcoords.py:87, patch drop
models/aggregators/point_aggregators.py:1222-1237, straight-through conf clamp :1422-1424) and of what torch autograd
derives from it.  Gradients are produced for points_embeding / points_conf / points_dir / points_color and every aggregator
parameter that takes part in the order-2 hybrid path (`color_branch` is constructed but unused, :542-553, and gets none).
Differentiable outputs: coarse_raycolor and conf_coefficient (the two the shipped loss terms read,
dev_scripts/w_scannet_etf/scene241.sh:146-151); every other output is returned detached.

All arithmetic runs in libhnr_hip.so; torch supplies device memory, streams and a handful of index bookkeeping ops.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import HnrError, CNT
from . import querier as Q
from .linear import PackedLinear, weight_grad
from .render import _i32, _f32, PointCloud


def drop_patch_rays(patch_size, patch_num, drop_ratio):
    """point_aggregators.py:14-23 -- rows (of the patch_num*patch_size square batch) whose image feature is dropped."""
    flag = np.zeros((patch_size * patch_num, patch_size * patch_num), dtype=bool)
    n = int(patch_num * patch_num * drop_ratio)
    row, col = n // patch_num, n % patch_num
    flag[0:row * patch_size, :] = True
    flag[row * patch_size:row * patch_size + patch_size, 0:col * patch_size] = True
    return np.where(flag.flatten())[0]


def ray_drop_flags(opt, ray_mask):
    """[R] uint8: rays whose merged image feature is zeroed at train time.  The reference indexes the drop pattern by
    VALID-ray row (`drop_ray_flag[ray_drop_positions, :]` over the R' compacted rays, :1225-1233), reproduced here."""
    if not (getattr(opt, "is_train", 0) and getattr(opt, "drop_ratio", 0) > 0 and getattr(opt, "random_position", 0) == 1):
        return None
    if not getattr(opt, "ray_points", 0) or getattr(opt, "drop_disturb_range", 0) != 0:
        raise HnrError("only ray-based image-feature drop with drop_disturb_range=0 is implemented (all 19 shipped scripts)")
    R = ray_mask.shape[0]
    m = ray_mask > 0
    if not getattr(opt, "drop_patch", 0):
        # `random.sample(range(R'), int(R' * drop_ratio))` over the valid-ray rows (:1231-1232; Python's RNG there, torch's here):
        # an exact-size uniform subset, chosen on the device without reading R' back
        keys = torch.rand(R, device=ray_mask.device).masked_fill(~m, 2.0)
        rank = torch.empty(R, dtype=torch.long, device=ray_mask.device)
        rank[torch.argsort(keys)] = torch.arange(R, device=ray_mask.device)
        n_drop = (m.sum() * float(opt.drop_ratio)).to(torch.long)           # int() truncation, like the reference
        return (m & (rank < n_drop)).to(torch.uint8).contiguous()
    ps, pn = int(opt.dilation_setup.split("_")[1]), int(opt.dilation_setup.split("_")[0])
    pos = drop_patch_rays(ps, pn, opt.drop_ratio)
    lut = torch.zeros(R + 1, dtype=torch.bool, device=ray_mask.device)
    pos = pos[pos < R]
    lut[torch.from_numpy(pos).to(ray_mask.device)] = True
    m = ray_mask > 0
    row = torch.cumsum(m.to(torch.int32), 0) - 1                      # valid-ray row of every ray
    return (m & lut[row.clamp(min=0).long()]).to(torch.uint8).contiguous()


class _Saved:
    pass


class TrainPath:
    """Forward (activations kept) and backward of one ray batch.  `renderer` is a HybridRenderer (grid / feature caches)."""

    def __init__(self, renderer):
        self.r = renderer
        self.agg = renderer.agg
        self.opt = renderer.opt
        self._pt_key, self._pt = None, None
        self._bbox0, self._bbox_key = None, None

    # transposed weights for the input-gradient GEMMs, cached with the forward pack
    def packed_t(self):
        pk = self.agg.packed()
        if "t" in pk:
            return pk["t"]
        a = self.agg
        tr = lambda w: PackedLinear(w.detach().t().contiguous(), None)
        w0 = a.aux_merge_weight_block[0].weight
        t = dict(
            b1_1=tr(a.block1[2].weight), b1_point=tr(a.block1[0].weight[:, :224]),
            b3_0=tr(a.block3[0].weight), b3_1=tr(a.block3[2].weight),
            cf=[tr(a.color_feature_branch[i].weight) for i in (0, 2, 4)],
            mw0_fd=tr(torch.cat([w0[:, :45], w0[:, 173:176]], dim=1)), mw0_cf=tr(w0[:, 45:173]),
            mw=[None, tr(a.aux_merge_weight_block[2].weight), tr(a.aux_merge_weight_block[4].weight)],
            mx=[tr(a.color_mixup_block[i].weight) for i in (0, 2, 4)],
        )
        pk["t"] = t
        return t

    # ---------------------------------------------------------------------------------------------- forward
    def forward(self, cloud, raydir, campos, camrot, bg_color, near, far, c2w_nearest, campos_nearest, intrinsic_nearest,
                images_nearest, frame_weight=None, tmid=None, ray_drop=None):
        L = _lib.lib()
        r, opt = self.r, self.opt
        g = _lib.require_gpu
        raydir = g(raydir, "raydir", torch.float32).reshape(-1, 3)
        campos = g(campos, "campos", torch.float32).reshape(3)
        camrot = g(camrot, "camrotc2w", torch.float32).reshape(3, 3)
        bg_color = g(bg_color, "bg_color", torch.float32).reshape(3)
        c2w_nearest = g(c2w_nearest, "c2w_nearest", torch.float32).reshape(-1, 4, 4)
        campos_nearest = g(campos_nearest, "campos_nearest", torch.float32).reshape(-1, 3)
        intrinsic_nearest = g(intrinsic_nearest, "intrinsic_nearest", torch.float32).reshape(3, 3)
        w2c_nearest = torch.inverse(c2w_nearest).contiguous()
        dev = raydir.device
        grid, hp = r.querier._grid_for(cloud.xyz[None])
        if tmid is None:
            tmid = r.querier._tmid_for(float(near), float(far), opt.z_depth_dim, raydir.shape[0], dev)
        qres = Q.march_query(grid, campos, raydir, tmid, opt.SR, opt.K, np.float32(hp[0] ** 2), opt.kernel_size, pad=True)
        pidx, loc_w, counts, work = qres["sample_pidx"], qres["sample_loc_w"], qres["counts"], qres["work"]
        R, SR, K = pidx.shape
        st, p = _lib.stream, _lib.ptr
        pk = self.agg.packed()
        sl = pk["slope"]
        S = _Saved()
        S.cloud, S.qres, S.raydir, S.campos, S.camrot, S.bg = cloud, qres, raydir, campos, camrot, bg_color
        S.w2c, S.Kn, S.campos_n = w2c_nearest, intrinsic_nearest, campos_nearest
        S.R, S.SR, S.K = R, SR, K
        decoded = torch.zeros((R, SR, 4), dtype=torch.float32, device=dev)
        S.decoded = decoded
        w_out = torch.zeros((R, SR, K), dtype=torch.float32, device=dev)
        c_out = cloud.conf[0].clamp(0.0001, 1.0).expand(R, SR, K).contiguous()
        S.w_out = w_out
        c = counts.cpu()
        n_valid, n_rows = int(c[CNT["SAMPLES_VALID"]]), int(c[CNT["NEIGHBOURS"]])
        S.n_valid, S.n_rows = n_valid, n_rows
        # ray_drop: explicit [R] flags (a rank's slice of the batch-wide drop pattern when the batch is sharded over GPUs)
        S.ray_drop = ray_drop_flags(opt, qres["ray_mask"]) if ray_drop is None else \
            (_lib.require_gpu(ray_drop, "ray_drop", torch.uint8).reshape(-1) & (qres["ray_mask"] > 0).to(torch.uint8)).contiguous()
        img = g(images_nearest, "images_nearest", torch.float32)
        if img.dim() == 5:
            img = img[0]
        S.img = img
        V, H, W = img.shape[0], img.shape[1], img.shape[2]
        S.V, S.H, S.W = V, H, W
        S.no_views = getattr(opt, "use_nearest", 4) == 0        # scene241.sh: image branch off, merged = 0 (point_aggregators.py:1257-1258)
        with torch.cuda.device(dev):
            if n_valid > 0 and not S.no_views:
                # reference-view pyramid (activations kept for the conv backward)
                fm = torch.empty((V, H, W, 48), dtype=torch.float32, device=dev)
                S.fm_scratch = torch.empty((max(int(L.hnr_image_features_scratch_elems(V, H, W)), 1),), dtype=torch.float32, device=dev)
                wp = (ctypes.c_void_p * 6)(*[t.data_ptr() for t in pk["conv_w"]])
                bp = (ctypes.c_void_p * 6)(*[t.data_ptr() for t in pk["conv_b"]])
                _lib.check(L.hnr_image_features(p(img), V, H, W, wp, bp, sl, p(S.fm_scratch), p(fm), st()), "hnr_image_features")
            if n_valid > 0:
                S.vs_item, S.vs_off, S.vs_cnt = _i32(n_valid, dev), _i32(n_valid, dev), _i32(n_valid, dev)
                scratch = _i32(2 * ((R * SR + 1023) // 1024) + 2, dev)
                overflow = torch.zeros(1, dtype=torch.int32, device=dev)
                _lib.check(L.hnr_sample_plan(p(work), p(pidx), p(counts), K, R * SR, p(S.vs_item), p(S.vs_off), p(S.vs_cnt), n_valid,
                                             n_rows, p(scratch), p(overflow), st()), "hnr_sample_plan")
                S.Xd, S.X3, S.wagg = _f32((n_rows, 64), dev), _f32((n_rows, 264), dev), _f32((n_rows,), dev)
                S.row_pid = _i32(n_rows, dev)
                _lib.check(L.hnr_gather_rows(p(cloud.xyz), p(cloud.emb), p(cloud.conf), p(cloud.dir), p(cloud.color), cloud.F,
                                             p(pidx), p(loc_w), p(raydir), p(campos), p(camrot), p(S.vs_item), p(S.vs_off), p(S.vs_cnt),
                                             p(counts), SR, K, n_valid, p(S.Xd), 64, p(S.X3), 264, p(S.wagg), p(w_out), p(c_out),
                                             p(S.row_pid), st()), "hnr_gather_rows")
                # the points this batch touches: only their rows of the per-point table are computed
                N = cloud.xyz.shape[0]
                cap_u = min(n_rows, N)
                S.uidx, S.ulist, S.row_u = _i32(N, dev), _i32(cap_u, dev), _i32(n_rows, dev)
                ucount = torch.zeros(1, dtype=torch.int32, device=dev)
                scan_scratch = _i32((N + 1023) // 1024 + 1, dev)
                _lib.check(L.hnr_unique_points(p(S.row_pid), n_rows, N, p(S.uidx), p(S.ulist), cap_u, p(S.row_u), p(ucount),
                                               p(scan_scratch), st()), "hnr_unique_points")
                S.U = int(ucount.item())
                Tu, S.E = self.agg.point_table(cloud.emb, ids=S.ulist, n_ids=S.U, want_rows=True)
                S.H1 = pk["b1_dist"].gather_add(S.Xd, Tu, S.row_u, act=True, slope=sl, K=60)
                if getattr(r, "dense", "f32") == "bf16x3":                           # the forward's 256-wide layers as in inference (exactly split bf16 operands)
                    ps = self.agg.packed_split()
                    l12, l30, l32 = ps["b1_2"], ps["b3_0"], ps["b3_2"]
                else:
                    l12, l30, l32 = pk["b1"][1], pk["b3"][0], pk["b3"][1]
                l12(S.H1, out=S.X3, act=True, slope=sl)                              # H2 into X3[:, :256]
                S.H3 = l30(S.X3, act=True, slope=sl, K=263)
                S.H4 = l32(S.H3, act=True, slope=sl)
                S.X5, S.sigma = _f32((n_valid, 280), dev), _f32((n_valid,), dev)
                _lib.check(L.hnr_ksum(p(S.H4), 256, p(S.wagg), p(pk["alpha_w"]), p(pk["alpha_b"]), p(S.vs_item), p(S.vs_off), p(S.vs_cnt),
                                      p(raydir), p(counts), SR, n_valid, p(S.X5), 280, p(S.sigma), st()), "hnr_ksum")
                S.T1 = pk["cf"][0](S.X5, act=True, slope=sl)
                S.T2 = pk["cf"][1](S.T1, act=True, slope=sl)
                S.CF = pk["cf"][2](S.T2, act=True, slope=sl)
                if S.no_views:
                    S.X7 = torch.zeros((n_valid, 92), dtype=torch.float32, device=dev)
                    S.X7[:, :45] = S.CF[:, :45]
                    S.fw = None
                else:
                    S.X6, S.vmask, S.row_s = _f32((V * n_valid, 48), dev), _f32((V * n_valid,), dev), _i32(V * n_valid, dev)
                    _lib.check(L.hnr_proj_rows(p(loc_w), p(S.vs_item), p(counts), p(w2c_nearest), p(intrinsic_nearest), p(campos),
                                               p(campos_nearest), p(fm), V, H, W, p(S.CF), 128, n_valid, p(S.X6), 48, p(S.vmask),
                                               p(S.row_s), st()), "hnr_proj_rows")
                    del fm
                    pre = pk["mw0_cf"](S.CF, act=False)
                    S.M1 = pk["mw0_fd"].gather_add(S.X6, pre, S.row_s, act=True, slope=sl)
                    S.M2 = pk["mw"][1](S.M1, act=True, slope=sl)
                    S.M3 = pk["mw"][2](S.M2, act=True, slope=sl)
                    S.X7 = _f32((n_valid, 92), dev)
                    S.fw = None if frame_weight is None else g(frame_weight, "frame_weight", torch.float32).reshape(-1)
                    _lib.check(L.hnr_merge(p(S.X6), 48, p(S.M3), 64, p(pk["mw_last_w"]), p(pk["mw_last_b"]), p(S.vmask),
                                           p(S.fw) if S.fw is not None else None, p(S.CF), 128, p(counts), V, n_valid, p(S.X7), 92,
                                           p(S.ray_drop) if S.ray_drop is not None else None, p(S.vs_item), SR, st()), "hnr_merge")
                S.Y1 = pk["mx"][0](S.X7, out=_f32((n_valid, 48), dev), act=True, slope=sl, K=90)
                S.Y2 = pk["mx"][1](S.Y1, out=_f32((n_valid, 48), dev), act=True, slope=sl, K=45)
                S.Y3 = pk["mx"][2](S.Y2, out=_f32((n_valid, 48), dev), act=False, K=45)
                _lib.check(L.hnr_final_color(p(S.Y3), 48, p(S.CF), 128, p(pk["fin_w"]), p(pk["fin_b"]), p(S.sigma), p(S.vs_item),
                                             p(counts), n_valid, p(decoded), st()), "hnr_final_color")
                if int(overflow.item()) != 0:
                    raise HnrError("hnr_sample_plan: row buffers too small (internal sizing error)")
        comp = r.composite(decoded, qres, campos, camrot, bg_color, want_blend=True)
        out = dict(comp)
        out.update(ray_mask=qres["ray_mask"], decoded=decoded, sample_pidx=pidx, sample_loc_w=loc_w, ray_nsamp=qres["ray_nsamp"],
                   counts=counts, weight=w_out, conf_coefficient=c_out)
        return out, S

    # ---------------------------------------------------------------------------------------------- backward
    def backward(self, S, g_raycolor, g_conf_out=None):
        """Returns (point grads dict, aggregator grads dict keyed by parameter name)."""
        L = _lib.lib()
        a, pk, t = self.agg, self.agg.packed(), self.packed_t()
        sl = pk["slope"]
        dev = S.raydir.device
        cloud, qres = S.cloud, S.qres
        R, SR, K, V = S.R, S.SR, S.K, S.V
        nS, M = S.n_valid, S.n_rows
        N = cloud.xyz.shape[0]
        st, p = _lib.stream, _lib.ptr
        z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
        pg = dict(points_embeding=z(N, cloud.F), points_conf=z(N), points_dir=z(N, 3), points_color=z(N, 3))
        ag = {}
        names = dict(a.named_parameters())
        for k, prm in names.items():
            skip = ("color_branch.", "learn_blur_kernel")         # unused head / modules the training shell runs itself
            if S.no_views:
                skip += ("aux_block_", "aux_merge_weight_block.")    # image branch off: like unused parameters in the reference, no gradient
            if not k.startswith(skip):
                ag[k] = torch.zeros_like(prm, dtype=torch.float32)
        if g_conf_out is not None:
            g_conf_out = _lib.require_gpu(g_conf_out, "grad conf_coefficient", torch.float32).reshape(R, SR, K)
            # empty slots read point 0 through the index clamp (neural_points.py:711): their gradient lands on conf[0]
            pg["points_conf"][0] += (g_conf_out * (qres["sample_pidx"] < 0)).sum()
        if nS == 0:
            return pg, ag
        g_raycolor = _lib.require_gpu(g_raycolor, "grad coarse_raycolor", torch.float32).reshape(R, 3)
        pidx, loc_w, counts = qres["sample_pidx"], qres["sample_loc_w"], qres["counts"]

        def lin_bwd(dZ, Xin, wname, bname, Nout, Kin, tw, prev=None, prev_cols=None, out=None, Kt=None):
            """dW/db of layer y = x W^T + b from dZ [rows, Nout] and its input Xin [rows, Kin]; returns dZ_prev =
            (dZ W) * LeakyReLU'(prev) (prev = stored activation of the producing layer) or the raw dZ W when prev is None."""
            weight_grad(dZ, Xin, Nout, Kin, dW=ag[wname], db=ag[bname] if bname else None, accumulate=False, want_bias=bname is not None)
            if tw is None:
                return None
            if prev is None:
                return tw(dZ, out=out, act=False, K=Kt if Kt is not None else Nout)
            return tw.side(dZ, prev, r_cols=prev_cols, r_mode=1, out=out, slope=sl, K=Kt if Kt is not None else Nout)

        with torch.cuda.device(dev):
            # 1. composite
            g_dec = _f32((R, SR, 4), dev)
            _lib.check(L.hnr_composite_bwd(p(S.decoded), p(loc_w), p(pidx), p(qres["ray_mask"]), None, p(S.campos), p(S.camrot),
                                           p(S.bg), R, SR, K, float(np.float32(self.opt.vsize[2])),
                                           int(getattr(self.opt, "raydist_mode_unit", 0) > 0), p(g_raycolor), p(g_dec), st()),
                       "hnr_composite_bwd")
            # 2. final colour
            gY3, gCF, g_sigma = _f32((nS, 48), dev), _f32((nS, 128), dev), _f32((nS,), dev)
            _lib.check(L.hnr_final_color_bwd(p(S.Y3), 48, p(S.CF), 128, p(pk["fin_w"]), p(pk["fin_b"]), p(S.vs_item), p(counts), nS,
                                             p(g_dec), p(gY3), 48, p(gCF), 128, p(g_sigma), p(ag["color_final_block.0.weight"]),
                                             p(ag["color_final_block.0.bias"]), st()), "hnr_final_color_bwd")
            # 3. mix-up block (last layer has no activation)
            dZ = lin_bwd(gY3, S.Y2, "color_mixup_block.4.weight", "color_mixup_block.4.bias", 45, 45, t["mx"][2], prev=S.Y2,
                         out=_f32((nS, 48), dev))
            dZ = lin_bwd(dZ, S.Y1, "color_mixup_block.2.weight", "color_mixup_block.2.bias", 45, 45, t["mx"][1], prev=S.Y1,
                         out=_f32((nS, 48), dev))
            gX7 = lin_bwd(dZ, S.X7, "color_mixup_block.0.weight", "color_mixup_block.0.bias", 45, 90, t["mx"][0], out=_f32((nS, 92), dev))
            if S.no_views:
                gCF[:, :45] += gX7[:, :45]                               # X7 = [colfeat[:45] | 0]
            else:
                # 4. merge
                gF, gZ3 = _f32((V * nS, 48), dev), _f32((V * nS, 64), dev)
                g_wl, g_bl = z(64), z(1)
                _lib.check(L.hnr_merge_bwd(p(S.X6), 48, p(S.M3), 64, p(pk["mw_last_w"]), p(pk["mw_last_b"]), p(S.vmask),
                                           p(S.fw) if S.fw is not None else None, p(counts), V, nS, sl,
                                           p(S.ray_drop) if S.ray_drop is not None else None, p(S.vs_item), SR, p(gX7), 92, p(gF), 48,
                                           p(gZ3), 64, p(gCF), 128, p(g_wl), p(g_bl), st()), "hnr_merge_bwd")
                ag["aux_merge_weight_block.6.weight"].copy_(g_wl.view(1, 64))
                ag["aux_merge_weight_block.6.bias"].copy_(g_bl)
                # 5. merge-weight MLP (first layer split: [imgfeat45 | ddir3] per row, colour feature once per sample)
                dZ = lin_bwd(gZ3, S.M2, "aux_merge_weight_block.4.weight", "aux_merge_weight_block.4.bias", 64, 64, t["mw"][2], prev=S.M2)
                dZ1 = lin_bwd(dZ, S.M1, "aux_merge_weight_block.2.weight", "aux_merge_weight_block.2.bias", 64, 64, t["mw"][1], prev=S.M1)
                G0 = ag["aux_merge_weight_block.0.weight"]                                # [64,176]
                gWfd, _ = weight_grad(dZ1, S.X6, 64, 48, want_bias=False)
                G0[:, :45].copy_(gWfd[:, :45])
                G0[:, 173:176].copy_(gWfd[:, 45:48])
                gpre = _f32((nS, 64), dev)
                _lib.check(L.hnr_sum_views(p(dZ1), 64, V, nS, nS, 64, p(gpre), 64, st()), "hnr_sum_views")
                weight_grad(gpre, S.CF, 64, 128, dW=G0[:, 45:173], db=ag["aux_merge_weight_block.0.bias"])
                gX6 = t["mw0_fd"](dZ1, act=False, K=64)                                   # [V*S,48]
                t["mw0_cf"].side(gpre, gCF, r_mode=0, out=gCF, act=False, K=64)           # gCF += gpre Wcf
                del dZ, dZ1, gZ3
                # 6. pixel gather + upsample + conv pyramid
                g_pyr = torch.zeros_like(S.fm_scratch)
                g_fm = z(V, S.H, S.W, 48)
                if self._bbox0 is None or self._bbox_key != (V, S.H, S.W, dev):
                    self._bbox0 = torch.tensor([[S.W, S.H, -1, -1]] * V, dtype=torch.int32, device=dev)
                    self._bbox_key = (V, S.H, S.W, dev)
                bbox = self._bbox0.clone()
                sb = int(L.hnr_sort_rows_scratch_bytes(V * nS))
                key_scratch, sort_scratch = _i32(3 * V * nS, dev), torch.empty((sb,), dtype=torch.uint8, device=dev)   # named: both must stay alive
                _lib.check(L.hnr_proj_rows_bwd(p(loc_w), p(S.vs_item), p(counts), p(S.w2c), p(S.Kn), V, S.H, S.W, nS, p(gF), 48, p(gX6), 48,
                                               p(g_fm), p(bbox), p(g_pyr), p(key_scratch), p(sort_scratch), sb, st()), "hnr_proj_rows_bwd")
                del g_fm
                conv_names = [("aux_block_s%d.%d" % (lvl, i)) for lvl in (1, 2, 3) for i in (0, 2)]
                wp = (ctypes.c_void_p * 6)(*[tt.data_ptr() for tt in pk["conv_w"]])
                gw = (ctypes.c_void_p * 6)(*[ag[n + ".weight"].data_ptr() for n in conv_names])
                gb = (ctypes.c_void_p * 6)(*[ag[n + ".bias"].data_ptr() for n in conv_names])
                _lib.check(L.hnr_image_features_bwd(p(S.img), V, S.H, S.W, wp, sl, p(S.fm_scratch), p(g_pyr), gw, gb, st()),
                           "hnr_image_features_bwd")
                del g_pyr, gF, gX6
            # 7. colour-feature branch
            _lib.check(L.hnr_dleaky(p(gCF), 128, p(S.CF), 128, nS, 128, sl, st()), "hnr_dleaky")
            dZ = lin_bwd(gCF, S.T2, "color_feature_branch.4.weight", "color_feature_branch.4.bias", 128, 128, t["cf"][2], prev=S.T2)
            dZ = lin_bwd(dZ, S.T1, "color_feature_branch.2.weight", "color_feature_branch.2.bias", 128, 128, t["cf"][1], prev=S.T1)
            gX5 = lin_bwd(dZ, S.X5, "color_feature_branch.0.weight", "color_feature_branch.0.bias", 128, 280, t["cf"][0])
            # 8. K-sum + alpha branch
            gZ4, g_wagg = _f32((M, 256), dev), _f32((M,), dev)
            g_aw, g_ab = z(256), z(1)
            _lib.check(L.hnr_ksum_bwd(p(S.H4), 256, p(S.wagg), p(pk["alpha_w"]), p(pk["alpha_b"]), p(S.vs_off), p(S.vs_cnt), p(counts), nS,
                                      p(gX5), 280, p(g_sigma), sl, p(gZ4), 256, p(g_wagg), p(g_aw), p(g_ab), st()), "hnr_ksum_bwd")
            ag["alpha_branch.0.weight"].copy_(g_aw.view(1, 256))
            ag["alpha_branch.0.bias"].copy_(g_ab)
            # 9. block3
            dZ3 = lin_bwd(gZ4, S.H3, "block3.2.weight", "block3.2.bias", 256, 256, t["b3_1"], prev=S.H3)
            del gZ4
            gX3 = lin_bwd(dZ3, S.X3, "block3.0.weight", "block3.0.bias", 256, 263, t["b3_0"], prev=S.X3, prev_cols=256,
                          out=_f32((M, 264), dev))
            del dZ3
            # rows -> touched point: sort the rows by touched-point index ONCE; both per-point reductions below then add a point's rows in
            # that fixed order (one wave per point, no atomics): gradients are bit-identical run to run
            sb = int(L.hnr_sort_rows_scratch_bytes(M))
            ks, perm, sort_scratch = _i32(M, dev), _i32(M, dev), torch.empty((sb,), dtype=torch.uint8, device=dev)
            _lib.check(L.hnr_sort_rows_by_key(p(S.row_u), M, p(ks), p(perm), p(sort_scratch), sb, st()), "hnr_sort_rows_by_key")
            # 10. point colour / direction / confidence
            G8, P8 = _f32((M, 8), dev), _f32((max(S.U, 1), 8), dev)
            _lib.check(L.hnr_gather_rows_bwd_rows(p(pidx), p(S.raydir), p(S.vs_item), p(S.vs_off), p(S.vs_cnt), p(counts), SR, K, nS, p(gX3), 264,
                                                  p(g_wagg), p(S.w_out), p(g_conf_out) if g_conf_out is not None else None, p(G8), st()),
                       "hnr_gather_rows_bwd_rows")
            _lib.check(L.hnr_segment_sum_rows_det(p(G8), 8, p(ks), p(perm), M, 8, S.U, None, p(P8), 8, 0, st()), "hnr_segment_sum_rows_det")
            _lib.check(L.hnr_point_small_grads(p(P8), p(S.ulist), S.U, p(pg["points_conf"]), p(pg["points_dir"]), p(pg["points_color"]), st()),
                       "hnr_point_small_grads")
            # 11. block1 (first layer split: 60 distance columns per row + per-point table)
            dZ2 = gX3[:, :256]
            dZ1 = lin_bwd(dZ2, S.H1, "block1.2.weight", "block1.2.bias", 256, 256, t["b1_1"], prev=S.H1)
            G1 = ag["block1.0.weight"]                                                # [256,284]
            weight_grad(dZ1, S.Xd, 256, 60, dW=G1[:, 224:284], db=ag["block1.0.bias"])
            gTu = _f32((max(S.U, 1), 256), dev)
            _lib.check(L.hnr_segment_sum_rows_det(p(dZ1), 256, p(ks), p(perm), M, 256, S.U, None, p(gTu), 256, 0, st()), "hnr_segment_sum_rows_det")
            if S.U > 0:
                gTu = gTu[:S.U]
                weight_grad(gTu, S.E, 256, 224, dW=G1[:, :224], want_bias=False)
                gE = t["b1_point"](gTu, act=False, K=256)                             # [U,224]
                _lib.check(L.hnr_point_rows_bwd(p(gE), 224, p(S.E), 224, p(S.ulist), S.U, cloud.F, p(pg["points_embeding"]), st()),
                           "hnr_point_rows_bwd")
        return pg, ag


class _RenderFn(torch.autograd.Function):
    """inputs: (path, static dict, emb, conf, dir, color, *aggregator parameters) -> (coarse_raycolor [R,3], conf_coefficient [R,SR,K],
    then detached extras).  The static dict carries the geometry / camera / image inputs (no gradient)."""

    @staticmethod
    def forward(ctx, path, static, emb, conf, pdir, color, *params):
        cloud = PointCloud(static["xyz"], emb, conf, pdir, color)
        out, S = path.forward(cloud, static["raydir"], static["campos"], static["camrot"], static["bg_color"], static["near"],
                              static["far"], static["c2w_nearest"], static["campos_nearest"], static["intrinsic_nearest"],
                              static["images_nearest"], frame_weight=static.get("frame_weight"), tmid=static.get("tmid"),
                              ray_drop=static.get("ray_drop"))
        ctx.path, ctx.S = path, S
        ctx.shapes = (emb.shape, conf.shape, pdir.shape, color.shape)
        ctx.param_names = static["param_names"]
        static["_out"] = out
        col, cc = out["coarse_raycolor"], out["conf_coefficient"]
        return col, cc

    @staticmethod
    def backward(ctx, g_col, g_cc):
        S = ctx.S
        if g_col is None:
            g_col = torch.zeros((S.R, 3), dtype=torch.float32, device=S.raydir.device)
        pg, ag = ctx.path.backward(S, g_col.contiguous(), None if g_cc is None else g_cc.contiguous())
        es, cs, ds, ks = ctx.shapes
        grads = [None, None, pg["points_embeding"].reshape(es), pg["points_conf"].reshape(cs), pg["points_dir"].reshape(ds),
                 pg["points_color"].reshape(ks)]
        for n in ctx.param_names:
            grads.append(ag.get(n))
        ctx.S = None
        return tuple(grads)


def render_train(path, aggregator, xyz, emb, conf, pdir, color, raydir, campos, camrot, bg_color, near, far, c2w_nearest,
                 campos_nearest, intrinsic_nearest, images_nearest, frame_weight=None, tmid=None, ray_drop=None):
    """Differentiable render of one ray batch.  emb/conf/pdir/color may be nn.Parameters (reference shapes [1,N,32], [1,N,1],
    [1,N,3], [1,N,3]); aggregator parameters receive gradients through the returned tensors.  Returns the output dict of
    TrainPath.forward with `coarse_raycolor` and `conf_coefficient` attached to the autograd graph."""
    names = [n for n, _ in aggregator.named_parameters()]
    params = [q for _, q in aggregator.named_parameters()]
    static = dict(xyz=xyz, raydir=raydir, campos=campos, camrot=camrot, bg_color=bg_color, near=near, far=far,
                  c2w_nearest=c2w_nearest, campos_nearest=campos_nearest, intrinsic_nearest=intrinsic_nearest,
                  images_nearest=images_nearest, frame_weight=frame_weight, tmid=tmid, ray_drop=ray_drop, param_names=names)
    col, cc = _RenderFn.apply(path, static, emb, conf, pdir, color, *params)
    out = dict(static.pop("_out"))
    out["coarse_raycolor"], out["conf_coefficient"] = col, cc
    return out
