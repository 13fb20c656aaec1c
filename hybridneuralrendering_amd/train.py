"""Training step of the path: TWO library calls per step (hnr_render_train_forward / hnr_render_train_backward, csrc/render_train.hip),
wrapped in one torch.autograd.Function so the reference's shell (losses, optimisers, schedulers) works unchanged.

Counterpart of NeuralPointsRayMarching.forward in train mode (/root/reference/models/neural_points_volumetric_model.py:257-427:
jittered depths models/neural_points/query_point_indices_worldcoords.py:87, patch drop
models/aggregators/point_aggregators.py:1222-1237, straight-through conf clamp :1422-1424) and of what torch autograd derives from it
(models/mvs_points_volumetric_model.py:111-131).  Gradients are produced for points_embeding / points_conf / points_dir / points_color
and every aggregator parameter that takes part in the order-2 hybrid path (`color_branch` is constructed but unused, :542-553, and gets
none).  Differentiable outputs: coarse_raycolor and conf_coefficient (the two the shipped loss terms read,
dev_scripts/w_scannet_etf/scene241.sh:146-151); every other output is returned detached.

Nothing here reads a device value: the library sizes every stage from device counters inside a workspace of `cap_samples` valid shading
samples (default R * SR, always enough); `TrainPath.check_status` reads the overflow word when the caller wants to (a host sync).
All arithmetic runs in libhnr_hip.so; torch supplies device memory and the stream.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import HnrError
from . import querier as Q
from .render import _f32, PointCloud


def drop_patch_rays(patch_size, patch_num, drop_ratio):
    """point_aggregators.py:14-23 -- rows (of the patch_num*patch_size square batch) whose image feature is dropped."""
    flag = np.zeros((patch_size * patch_num, patch_size * patch_num), dtype=bool)
    n = int(patch_num * patch_num * drop_ratio)
    row, col = n // patch_num, n % patch_num
    flag[0:row * patch_size, :] = True
    flag[row * patch_size:row * patch_size + patch_size, 0:col * patch_size] = True
    return np.where(flag.flatten())[0]


def _drop_mode(opt):
    """None: no image-feature drop; "patch": the deterministic patch pattern; "random": an exact-size random subset of the valid rays."""
    if not (getattr(opt, "is_train", 0) and getattr(opt, "drop_ratio", 0) > 0 and getattr(opt, "random_position", 0) == 1):
        return None
    if not getattr(opt, "ray_points", 0) or getattr(opt, "drop_disturb_range", 0) != 0:
        raise HnrError("only ray-based image-feature drop with drop_disturb_range=0 is implemented (all 19 shipped scripts)")
    return "patch" if getattr(opt, "drop_patch", 0) else "random"


def drop_lut(opt, R, device):
    """[R] uint8 indexed by VALID-ray row: the reference applies `drop_ray_flag[ray_drop_positions, :]` to the R' compacted rays (:1225-1233),
    so the library looks the pattern up by the number of valid rays before a ray (hnr_render_train_forward's d_drop_lut)."""
    ps, pn = int(opt.dilation_setup.split("_")[1]), int(opt.dilation_setup.split("_")[0])
    pos = drop_patch_rays(ps, pn, opt.drop_ratio)
    lut = np.zeros((R,), dtype=np.uint8)
    lut[pos[pos < R]] = 1
    return torch.from_numpy(lut).to(device)


def ray_drop_flags(opt, ray_mask):
    """[R] uint8: rays whose merged image feature is zeroed at train time, from a ray mask (host-side form of what the library does with
    d_drop_lut; also the random mode's flag builder).  The reference indexes the drop pattern by VALID-ray row."""
    mode = _drop_mode(opt)
    if mode is None:
        return None
    R = ray_mask.shape[0]
    m = ray_mask > 0
    if mode == "random":
        # `random.sample(range(R'), int(R' * drop_ratio))` over the valid-ray rows (:1231-1232; Python's RNG there, torch's here):
        # an exact-size uniform subset, chosen on the device without reading R' back
        keys = torch.rand(R, device=ray_mask.device).masked_fill(~m, 2.0)
        rank = torch.empty(R, dtype=torch.long, device=ray_mask.device)
        rank[torch.argsort(keys)] = torch.arange(R, device=ray_mask.device)
        n_drop = (m.sum() * float(opt.drop_ratio)).to(torch.long)           # int() truncation, like the reference
        return (m & (rank < n_drop)).to(torch.uint8).contiguous()
    lut = drop_lut(opt, R, ray_mask.device).to(torch.bool)
    row = torch.cumsum(m.to(torch.int32), 0) - 1                      # valid-ray row of every ray
    return (m & lut[row.clamp(min=0).long()]).to(torch.uint8).contiguous()


# parameter name -> (field of hnr_train_weights, index or None)
def _weight_slots():
    slots = {}
    for nm, fld in (("block1.0", "block1_0"), ("block1.2", "block1_2"), ("block3.0", "block3_0"), ("block3.2", "block3_2"), ("alpha_branch.0", "alpha"),
                    ("color_final_block.0", "fin")):
        slots[nm + ".weight"] = (fld + "_w", None)
        slots[nm + ".bias"] = (fld + "_b", None)
    for i, l in enumerate((0, 2, 4)):
        slots["color_feature_branch.%d.weight" % l] = ("cf_w", i); slots["color_feature_branch.%d.bias" % l] = ("cf_b", i)
        slots["color_mixup_block.%d.weight" % l] = ("mx_w", i); slots["color_mixup_block.%d.bias" % l] = ("mx_b", i)
    for i, l in enumerate((0, 2, 4, 6)):
        slots["aux_merge_weight_block.%d.weight" % l] = ("mw_w", i); slots["aux_merge_weight_block.%d.bias" % l] = ("mw_b", i)
    for lvl in (1, 2, 3):
        for j, l in enumerate((0, 2)):
            slots["aux_block_s%d.%d.weight" % (lvl, l)] = ("conv_w", 2 * (lvl - 1) + j); slots["aux_block_s%d.%d.bias" % (lvl, l)] = ("conv_b", 2 * (lvl - 1) + j)
    return slots


_SLOTS = _weight_slots()


def _fill_weights(tensors):
    """hnr_train_weights from {parameter name: contiguous fp32 GPU tensor}."""
    w = _lib.TrainWeights()
    for name, (fld, idx) in _SLOTS.items():
        t = tensors.get(name)
        p = ctypes.c_void_p(t.data_ptr()) if t is not None else None
        if idx is None:
            setattr(w, fld, p)
        else:
            getattr(w, fld)[idx] = p
    return w


class _Saved:
    pass


class TrainPath:
    """Forward (activations kept in the library's workspace) and backward of one ray batch.  `renderer` is a HybridRenderer (grid cache)."""

    def __init__(self, renderer):
        self.r = renderer
        self.agg = renderer.agg
        self.opt = renderer.opt
        self.cap_samples = None            # valid-sample capacity of the workspace (None: R * SR, the worst case)
        self.timers = None                 # a dict: the library records HIP events at its stage boundaries (profiling; read them after a synchronise)
        self.workspace = None              # tools: a caller-owned uint8 tensor every forward uses instead of a fresh torch.empty (one step in flight at a time)
        self._lut = {}
        self._wcache = {}
        self._ocache = {}
        # reuse_outputs: every step writes its outputs and gradients into the SAME tensors (allocated once per batch shape) -- what a training loop wants
        # (the optimiser has consumed a step's gradients before the next step runs) and ~0.15 ms less Python between the launches of a step; off by default
        # because a caller that keeps two steps' results alive would see the first overwritten
        self.reuse_outputs = False

    # ---------------------------------------------------------------------------------------------- forward
    def forward(self, cloud, raydir, campos, camrot, bg_color, near, far, c2w_nearest, campos_nearest, intrinsic_nearest,
                images_nearest, frame_weight=None, tmid=None, ray_drop=None, w2c_nearest=None):
        """w2c_nearest [V,4,4]: inverse(c2w_nearest) computed by the caller (torch.inverse may synchronise the host: a captured step passes it in)."""
        L = _lib.lib()
        r, opt = self.r, self.opt
        g, p = _lib.require_gpu, _lib.ptr
        raydir = g(raydir, "raydir", torch.float32).reshape(-1, 3)
        campos = g(campos, "campos", torch.float32).reshape(3)
        camrot = g(camrot, "camrotc2w", torch.float32).reshape(3, 3)
        bg_color = g(bg_color, "bg_color", torch.float32).reshape(3)
        c2w_nearest = g(c2w_nearest, "c2w_nearest", torch.float32).reshape(-1, 4, 4)
        campos_nearest = g(campos_nearest, "campos_nearest", torch.float32).reshape(-1, 3)
        intrinsic_nearest = g(intrinsic_nearest, "intrinsic_nearest", torch.float32).reshape(3, 3)
        dev = raydir.device
        if int(opt.K) != 8 or cloud.F != 32:
            raise HnrError("the training path is built for K = 8 neighbours and 32 point features (got K=%d, F=%d)" % (opt.K, cloud.F))
        R, SR, K = raydir.shape[0], int(opt.SR), 8
        if R == 0:
            raise HnrError("render_train: empty ray batch")
        grid, hp = r.querier._grid_for(cloud.xyz[None])
        if tmid is None:
            tmid = r.querier._tmid_for(float(near), float(far), opt.z_depth_dim, R, dev)
        tmid = g(tmid, "tmid", torch.float32)
        no_views = getattr(opt, "use_nearest", 4) == 0        # scene241.sh: image branch off, merged = 0 (point_aggregators.py:1257-1258)
        img = g(images_nearest, "images_nearest", torch.float32)
        if img.dim() == 5:
            img = img[0]
        V, H, W = (0, 0, 0) if no_views else (int(img.shape[0]), int(img.shape[1]), int(img.shape[2]))
        S = _Saved()
        prm = _lib.TrainParams()
        prm.R, prm.SR, prm.K, prm.D = R, SR, K, int(tmid.shape[-1])
        prm.tmid_stride = 0 if tmid.dim() == 1 else int(tmid.shape[1])
        for i in range(3):
            prm.kernel_size[i] = int(opt.kernel_size[i])
        prm.radius2, prm.vsize_z = float(np.float32(hp[0] ** 2)), float(np.float32(opt.vsize[2]))
        prm.raydist_mode_unit = int(getattr(opt, "raydist_mode_unit", 0) > 0)
        prm.V, prm.H, prm.W = V, H, W
        prm.n_points = int(cloud.xyz.shape[0])
        prm.cap_samples = int(self.cap_samples) if self.cap_samples else R * SR
        prm.knn_order = 0
        prm.slope = float(self.agg.block1[1].negative_slope)
        nbytes = int(L.hnr_render_train_workspace_bytes(ctypes.byref(prm)))
        if nbytes < 0:
            raise HnrError("hnr_render_train_workspace_bytes: %s" % L.hnr_last_error().decode("utf-8", "replace"))
        capturing = torch.cuda.is_current_stream_capturing()
        free, _total = (1 << 62, 0) if capturing else torch.cuda.mem_get_info(dev)
        cached = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
        if nbytes > 0.9 * (free + cached):
            raise HnrError("render_train: the workspace for %d rays x SR %d (%.1f GB) does not fit the %.1f GB that are free; set TrainPath.cap_samples to the "
                           "number of valid shading samples a batch can produce" % (R, SR, nbytes / 1e9, (free + cached) / 1e9))
        if self.reuse_outputs and self.workspace is None:
            self.workspace = torch.empty((nbytes + 256,), dtype=torch.uint8, device=dev)     # (one step in flight at a time: the same workspace every step)
        ws = self.workspace if (self.workspace is not None and self.workspace.numel() >= nbytes + 256) else torch.empty((nbytes + 256,), dtype=torch.uint8, device=dev)
        off = (-ws.data_ptr()) % 256
        # parameters as contiguous fp32 tensors under the reference's names (views of the nn.Parameters); the tensors and the ctypes block are kept
        # while the parameters stay where they are (an optimiser updates them in place): ~0.1 ms of Python per step otherwise
        pkey = tuple((n, q.data_ptr()) for n, q in self.agg.named_parameters() if n in _SLOTS)
        if self._wcache.get("key") != pkey:
            wt = {n: g(q.detach(), n, torch.float32) for n, q in self.agg.named_parameters() if n in _SLOTS}
            self._wcache = dict(key=pkey, wt=wt, weights=_fill_weights(wt))
        wt = self._wcache["wt"]
        S.wt, S.weights = wt, self._wcache["weights"]
        S.cloud_t = (cloud.xyz, cloud.emb, cloud.conf, cloud.dir, cloud.color)
        S.cl = _lib.TrainCloud(p(cloud.xyz), p(cloud.emb), p(cloud.conf), p(cloud.dir), p(cloud.color))
        S.cam_t = (campos, camrot, raydir, tmid, bg_color)
        S.cam = _lib.RenderCamera(p(campos), p(camrot), p(raydir), p(tmid), p(bg_color))
        S.vw, S.vw_t = None, None
        if V > 0:
            if w2c_nearest is not None:
                w2c = g(w2c_nearest, "w2c_nearest", torch.float32).reshape(-1, 4, 4)
            else:
                w2c = torch.inverse(c2w_nearest).contiguous()      # 4x4 plumbing op (neural_points_volumetric_model.py:250)
            fw = None if frame_weight is None else g(frame_weight, "frame_weight_nearest", torch.float32).reshape(-1)
            if fw is not None and fw.numel() != V:
                raise HnrError("frame_weight_nearest must hold one weight per reference view (%d), got %d values -- the item's scalar loss weight "
                               "`frame_weight` is a different input (train_step(frame_weight=...))" % (V, fw.numel()))
            S.vw_t = (w2c, intrinsic_nearest, campos_nearest, img, fw)
            S.vw = _lib.TrainViews(p(w2c), p(intrinsic_nearest), p(campos_nearest), p(img), p(fw) if fw is not None else None)
        lut = flags = None
        mode = _drop_mode(opt)
        if ray_drop is not None:
            # explicit [R] flags (a rank's slice of the batch-wide drop pattern when the batch is sharded over GPUs)
            flags = g(ray_drop, "ray_drop", torch.uint8).reshape(-1)
        elif mode == "patch" and V > 0:
            key = (R, str(opt.dilation_setup), float(opt.drop_ratio), str(dev))
            if self._lut.get("key") != key:                      # (a host -> device copy: once per batch shape, never inside a captured step)
                self._lut = dict(key=key, lut=drop_lut(opt, R, dev))
            lut = self._lut["lut"]
        elif mode == "random" and V > 0:
            # the random subset needs the ray mask first: one extra query launch (no host read), then explicit flags
            q0 = Q.march_query(grid, campos, raydir, tmid, SR, K, np.float32(hp[0] ** 2), opt.kernel_size, pad=True)
            flags = ray_drop_flags(opt, q0["ray_mask"])
        okey = ("fwd", R, SR, K, str(dev))
        if self.reuse_outputs and okey in self._ocache:
            col, opa, isbg, bw, mask, decoded, pidx, loc, nsamp, counts, status, w_out, c_out = self._ocache[okey]
        else:
            col, opa, isbg, bw = _f32((R, 3), dev), _f32((R, SR), dev), _f32((R,), dev), _f32((R, SR), dev)
            mask = torch.empty((R,), dtype=torch.int8, device=dev)
            decoded = _f32((R, SR, 4), dev)
            pidx = torch.empty((R, SR, K), dtype=torch.int32, device=dev)
            loc = _f32((R, SR, 3), dev)
            nsamp = torch.empty((R,), dtype=torch.int32, device=dev)
            counts = torch.empty((_lib.NCOUNTS,), dtype=torch.int64, device=dev)
            status = torch.empty((2,), dtype=torch.int32, device=dev)
            w_out, c_out = _f32((R, SR, K), dev), _f32((R, SR, K), dev)
            if self.reuse_outputs:
                self._ocache[okey] = (col, opa, isbg, bw, mask, decoded, pidx, loc, nsamp, counts, status, w_out, c_out)
        S.out = _lib.RenderOutputs(p(col), p(opa), p(isbg), p(bw), p(mask), p(decoded), p(pidx), p(loc), p(nsamp), p(counts), p(status), p(w_out), p(c_out), None)
        S.prm, S.ws, S.ws_ptr, S.nbytes, S.dev = prm, ws, ctypes.c_void_p(ws.data_ptr() + off), nbytes, dev
        if self.timers is not None:
            ev = _lib.StageEvents(_lib.TRAIN_FWD_STAGES)
            self.timers.setdefault("fwd", []).append(ev)
            S.out.stage_events = ev.arr
        S.R, S.SR, S.K, S.V, S.no_views = R, SR, K, V, no_views
        with torch.cuda.device(dev):
            _lib.check(L.hnr_render_train_forward(grid.handle, ctypes.byref(prm), ctypes.byref(S.cl), ctypes.byref(S.weights), ctypes.byref(S.cam),
                                                  ctypes.byref(S.vw) if S.vw is not None else None, p(lut) if lut is not None else None,
                                                  p(flags) if flags is not None else None, S.ws_ptr, nbytes, ctypes.byref(S.out), _lib.stream()),
                       "hnr_render_train_forward")
        S.keep = (lut, flags, grid)
        out = dict(coarse_raycolor=col, coarse_point_opacity=opa, coarse_is_background=isbg, blend_weight=bw, ray_mask=mask, decoded=decoded,
                   sample_pidx=pidx, sample_loc_w=loc, ray_nsamp=nsamp, counts=counts, status=status, weight=w_out, conf_coefficient=c_out)
        S.outs = out
        self.last_step = (prm, nbytes)     # tools (bisect_train_forward.py): the step's parameter block and workspace size
        return out, S

    @staticmethod
    def touched_points(S):
        """(ids int32 [capacity] ascending, count int64 [1]) -- views INTO the step's workspace (valid until the next forward on it): the points the
        batch referenced, as the forward call left them on the device (hnr_render_train_touched).  No sort, no host read."""
        L = _lib.lib()
        ids, cnt, cap = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64()
        _lib.check(L.hnr_render_train_touched(ctypes.byref(S.prm), S.ws_ptr, S.nbytes, ctypes.byref(ids), ctypes.byref(cnt), ctypes.byref(cap)),
                   "hnr_render_train_touched")
        base = S.ws.data_ptr()
        o_ids, o_cnt = ids.value - base, cnt.value - base
        return S.ws[o_ids:o_ids + 4 * cap.value].view(torch.int32), S.ws[o_cnt:o_cnt + 8].view(torch.int64)

    @staticmethod
    def check_status(out):
        """Reads the forward's status word (a host synchronisation): raises when the workspace capacity was exceeded."""
        st = out["status"].cpu()
        if int(st[0]) != 0:
            raise HnrError("render_train: %d valid shading samples exceed the workspace capacity (TrainPath.cap_samples); the extra ones were dropped" % int(st[1]))

    # ---------------------------------------------------------------------------------------------- backward
    def backward(self, S, g_raycolor, g_conf_out=None):
        """Returns (point grads dict, aggregator grads dict keyed by parameter name)."""
        L = _lib.lib()
        dev, p = S.dev, _lib.ptr
        N = int(S.cloud_t[0].shape[0])
        g_raycolor = _lib.require_gpu(g_raycolor, "grad coarse_raycolor", torch.float32).reshape(S.R, 3)
        if g_conf_out is not None:
            g_conf_out = _lib.require_gpu(g_conf_out, "grad conf_coefficient", torch.float32).reshape(S.R, S.SR, S.K)
        bkey = ("bwd", N, S.no_views, id(S.wt), str(dev))
        if self.reuse_outputs and bkey in self._ocache:
            pg, ag, gw, cg, flat, payload = self._ocache[bkey]
        else:
            pg = dict(points_embeding=_f32((N, 32), dev), points_conf=_f32((N,), dev), points_dir=_f32((N, 3), dev), points_color=_f32((N, 3), dev))
            skip = ()
            if S.no_views:
                skip = ("aux_block_", "aux_merge_weight_block.")     # image branch off: like unused parameters in the reference, no gradient
            names = [n for n in S.wt if not n.startswith(skip)]
            sizes = [int(S.wt[n].numel()) for n in names]
            offs = np.concatenate([[0], np.cumsum([(s + 63) // 64 * 64 for s in sizes])])
            flat = _f32((int(offs[-1]) + 64,), dev)               # one buffer for all weight gradients (256-byte aligned slices) + a spare tail (parallel.allreduce_weight_grads)
            ag = {n: flat[int(offs[i]):int(offs[i]) + sizes[i]].view(S.wt[n].shape) for i, n in enumerate(names)}
            gw = _fill_weights(ag)
            payload = int(offs[-1])
            cg = _lib.TrainCloudGrads(p(pg["points_embeding"]), p(pg["points_conf"]), p(pg["points_dir"]), p(pg["points_color"]))
            if self.reuse_outputs:
                self._ocache[bkey] = (pg, ag, gw, cg, flat, payload)
        S.flat, S.flat_payload = flat, payload                   # all weight gradients as ONE buffer: one all-reduce when the batch is sharded over ranks
        S.out.stage_events = None
        if self.timers is not None:
            ev = _lib.StageEvents(_lib.TRAIN_BWD_STAGES)
            self.timers.setdefault("bwd", []).append(ev)
            S.out.stage_events = ev.arr
        with torch.cuda.device(dev):
            _lib.check(L.hnr_render_train_backward(ctypes.byref(S.prm), ctypes.byref(S.cl), ctypes.byref(S.weights), ctypes.byref(S.cam),
                                                   ctypes.byref(S.vw) if S.vw is not None else None, S.ws_ptr, S.nbytes, ctypes.byref(S.out), p(g_raycolor),
                                                   p(g_conf_out) if g_conf_out is not None else None, ctypes.byref(cg), ctypes.byref(gw), _lib.stream()),
                       "hnr_render_train_backward")
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
            g_col = torch.zeros((S.R, 3), dtype=torch.float32, device=S.dev)
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


def _queue_step(path, cloud, raydir, campos, camrot, bg_color, near, far, c2w_nearest, campos_nearest, intrinsic_nearest, images_nearest, gt_image,
                zero_epsilon, w_color, w_zero_one, frame_weight, tmid, ray_drop, frame_weight_nearest, blur, w2c_nearest, device_frame_weight=None):
    """The launches of one step, queued back to back on the current stream: forward -> [blur module] -> loss kernels -> [blur module backward] ->
    backward.  Nothing is read back; shared by train_step (eager) and CapturedTrainStep (inside a hipGraph capture)."""
    from .losses import shipped_loss_grads
    from .blur import blur_select, blur_select_bwd
    out, S = path.forward(cloud, raydir, campos, camrot, bg_color, near, far, c2w_nearest, campos_nearest, intrinsic_nearest, images_nearest,
                          frame_weight=frame_weight_nearest, tmid=tmid, ray_drop=ray_drop, w2c_nearest=w2c_nearest)
    col = out["coarse_raycolor"]
    sel = None
    if blur is not None:
        # add_blur_sim=1 (models/mvs_points_volumetric_model.py:145-146 -> base_rendering_model.py:677-745): per patch, the pre-defined kernel whose
        # blurred render is closest to the ground truth replaces the render before the losses
        kernels, pn, ps = blur
        col, sel, kk = blur_select(col, gt_image, kernels, pn, ps)
    parts, g_col, g_cc = shipped_loss_grads(col, out["conf_coefficient"], gt_image, out["ray_mask"], zero_epsilon, w_color, w_zero_one, frame_weight,
                                            conf_rows=True, device_frame_weight=device_frame_weight)
    if blur is not None:
        g_col = blur_select_bwd(g_col, kk, sel, pn, ps)
    pg, ag = path.backward(S, g_col, g_cc)
    out = dict(out)
    out["loss"] = parts
    if blur is not None:
        out["blurred_raycolor"], out["blur_select"] = col, sel
    return out, S, pg, ag


def _blur_arg(blur_kernels, patch_num, patch_size, patch_layout):
    if blur_kernels is None:
        return None
    if patch_layout not in ("grid", "patch_major") or not patch_num or not patch_size:
        raise HnrError("train_step: blur_kernels need patch_num, patch_size and patch_layout 'grid' | 'patch_major'")
    return (blur_kernels, -int(patch_num) if patch_layout == "patch_major" else int(patch_num), int(patch_size))


def _assign_grads(aggregator, emb, conf, pdir, color, pg, ag, accumulate=True, cached=False):
    """Sets / adds to the .grad fields as autograd would.  `cached`: pg / ag are buffers the NEXT step overwrites in place (TrainPath.reuse_outputs,
    CapturedTrainStep): with accumulate a .grad never aliases them (first assignment clones, later ones add in place into the clone), so
    optimizer.zero_grad(set_to_none=False) and gradient accumulation see autograd's semantics; accumulate=False hands out the buffers themselves."""
    def put(t, g):
        if not (isinstance(t, torch.Tensor) and t.requires_grad):
            return
        g = g.reshape(t.shape)
        if not accumulate:
            t.grad = g
        elif t.grad is None:
            t.grad = g.clone() if cached else g
        elif t.grad.data_ptr() == g.data_ptr():
            raise HnrError("train_step: .grad of a parameter IS the step's own gradient buffer (an earlier step assigned it with accumulate_grads=False), "
                           "so its previous value is already overwritten; set the gradients to None or keep accumulate_grads=False")
        else:
            t.grad.add_(g)
    put(emb, pg["points_embeding"]); put(conf, pg["points_conf"]); put(pdir, pg["points_dir"]); put(color, pg["points_color"])
    for n, q in aggregator.named_parameters():
        if n in ag:
            put(q, ag[n])


def train_step(path, aggregator, xyz, emb, conf, pdir, color, raydir, campos, camrot, bg_color, near, far, c2w_nearest, campos_nearest,
               intrinsic_nearest, images_nearest, gt_image, zero_epsilon=1e-3, w_color=1.0, w_zero_one=1e-4, frame_weight=None, tmid=None,
               ray_drop=None, assign_grads=True, frame_weight_nearest=None, blur_kernels=None, patch_num=None, patch_size=None, patch_layout="grid",
               w2c_nearest=None, accumulate_grads=True):
    """forward -> [blur module] -> shipped loss terms -> backward of one ray batch as groups of library launches queued back to back: no autograd
    graph, no masked copies, nothing read back to the host -- the body of the reference's optimize_parameters before its optimizer steps
    (models/neural_points_volumetric_model.py:202-214: self.forward(); loss_total.backward(), with compute_losses of
    models/base_rendering_model.py:1060-1245 in between).  Same arithmetic as render_train + losses.shipped_loss + loss.backward().

    blur_kernels [1,N,ks,ks] / [N,ks,ks] (the item's `blur_kernels`, add_blur_sim=1) with patch_num / patch_size / patch_layout: the blur-handling
    module between the render and the losses (models/mvs_points_volumetric_model.py:145-146, base_rendering_model.py:677-745) -- hnr_blur_select before
    the loss kernels, hnr_blur_select_bwd behind them; patch_layout="patch_major": the batch is `patch_num` whole patches packed (patch, y, x), a rank's
    share of a patch-sharded batch (parallel.shard_patches).  Same arithmetic as render_train + blur.blur_update_output + the loss + loss.backward().

    The reference has TWO frame-weight inputs and so has this call: `frame_weight` is the dataset item's scalar that multiplies loss_total
    (models/base_rendering_model.py:1205; a Python float or a CPU tensor -- converted once, no device read in the step) and
    `frame_weight_nearest` [V] / [1,V] the per-reference-view weights of the image-feature merge under downweight_blurry_feats
    (models/aggregators/point_aggregators.py:1203), a device tensor that only the forward / backward calls read.

    emb/conf/pdir/color and the aggregator's parameters are read as they are; with assign_grads their .grad fields are set (or added to,
    as autograd does -- also under TrainPath.reuse_outputs, where a .grad is then a private copy of the step's buffer, never the buffer the next
    step overwrites; accumulate_grads=False assigns the step's own buffers instead: no copy, valid until the next step).  Returns (outputs dict with `loss` = {total, colour MSE, zero-one mean, valid rays} on the device and `_saved` = the step's
    state for TrainPath.touched_points / the flat weight-gradient buffer, point grads dict, aggregator grads dict keyed by parameter name)."""
    cloud = PointCloud(xyz, emb.detach(), conf.detach(), pdir.detach(), color.detach())
    out, S, pg, ag = _queue_step(path, cloud, raydir, campos, camrot, bg_color, near, far, c2w_nearest, campos_nearest, intrinsic_nearest, images_nearest,
                                 gt_image, zero_epsilon, w_color, w_zero_one, frame_weight, tmid, ray_drop, frame_weight_nearest,
                                 _blur_arg(blur_kernels, patch_num, patch_size, patch_layout), w2c_nearest)
    out["_saved"] = S
    if assign_grads:
        _assign_grads(aggregator, emb, conf, pdir, color, pg, ag, accumulate=bool(accumulate_grads), cached=bool(path.reuse_outputs))
    return out, pg, ag


class CapturedTrainStep:
    """train_step captured ONCE in a hipGraph (torch.cuda.CUDAGraph over the HIP graph API) and replayed per step.

    One step is ~150 launches of 10 - 50 us from three queues (the caller's stream and the library's two side streams, forked and joined with events
    inside the two library calls: stream capture follows them, so the graph keeps the three branches); replaying it removes the host's launch cost and
    the gaps between dependent launches -- what the reference pays per step in Python dispatch (models/neural_points_volumetric_model.py:202-214).
    Every launch size of the step is capacity-fixed and reads its true size from device counters, so ONE graph serves every batch of the same shape.

    Inputs live in static device buffers (`self.inputs`: raydir, campos, camrot, bg_color, c2w_nearest, w2c_nearest, campos_nearest, intrinsic_nearest,
    images_nearest, gt_image and, when given at construction, tmid, ray_drop, frame_weight_nearest, blur_kernels; `frame_weight` = one float);
    `step(**tensors)` copies what it is given into them and replays.  tmid=None at construction: the jittered depth tables are drawn INSIDE the graph
    (torch.rand under capture advances the registered generator state on every replay: a fresh jitter per step, as at
    models/neural_points/query_point_indices_worldcoords.py:87).  The weights and the point buffers are read through the pointers they had at capture:
    optimisers that update in place are fine; after prune / grow (new buffers) build a new CapturedTrainStep.  Outputs and gradients are static
    tensors overwritten by every replay."""

    def __init__(self, path, aggregator, xyz, emb, conf, pdir, color, sample, near, far, zero_epsilon=1e-3, w_color=1.0, w_zero_one=1e-4,
                 patch_num=None, patch_size=None, patch_layout="grid", warmup=2):
        """sample: dict of example input tensors (see the class docstring) that fixes shapes and optional inputs."""
        self.path, self.agg = path, aggregator
        self.leaves = (emb, conf, pdir, color)
        dev = xyz.device
        req = ("raydir", "campos", "camrot", "bg_color", "c2w_nearest", "campos_nearest", "intrinsic_nearest", "images_nearest", "gt_image")
        for k in req:
            if k not in sample:
                raise HnrError("CapturedTrainStep: sample input %r is missing" % k)
        st = {k: _lib.require_gpu(v, k).clone() for k, v in sample.items() if isinstance(v, torch.Tensor) and k != "frame_weight"}
        if "w2c_nearest" not in st:
            st["w2c_nearest"] = torch.inverse(st["c2w_nearest"].reshape(-1, 4, 4)).contiguous()
        fw0 = sample.get("frame_weight")
        fw0 = 1.0 if fw0 is None else (float(fw0.reshape(-1)[0].item()) if isinstance(fw0, torch.Tensor) else float(fw0))   # (construction time: a host read is fine)
        st["frame_weight"] = torch.full((1,), fw0, dtype=torch.float32, device=dev)
        self.inputs = st
        blur = _blur_arg(st.get("blur_kernels"), patch_num, patch_size, patch_layout)
        cloud = PointCloud(xyz, emb.detach(), conf.detach(), pdir.detach(), color.detach())

        def body():
            return _queue_step(path, cloud, st["raydir"], st["campos"], st["camrot"], st["bg_color"], near, far, st["c2w_nearest"], st["campos_nearest"],
                               st["intrinsic_nearest"], st["images_nearest"], st["gt_image"], zero_epsilon, w_color, w_zero_one, None, st.get("tmid"),
                               st.get("ray_drop"), st.get("frame_weight_nearest"), blur, st["w2c_nearest"], device_frame_weight=st["frame_weight"])
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, int(warmup))):       # lazy one-time work of the library (kernel attributes, side streams, the grid) happens here, not under capture
                out, _S, _pg, _ag = body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        TrainPath.check_status(out)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
            self.out, self.S, self.pg, self.ag = body()
        self.out["_saved"] = self.S

    def step(self, assign_grads=True, frame_weight=None, **tensors):
        """Copies the given inputs into the static buffers (device-to-device, no synchronisation), replays the graph, returns (out, pg, ag) --
        the same static tensors every time."""
        for k, v in tensors.items():
            if k not in self.inputs:
                raise HnrError("CapturedTrainStep.step: %r was not an input at capture (have %s)" % (k, sorted(self.inputs)))
            self.inputs[k].copy_(v.reshape(self.inputs[k].shape), non_blocking=True)
        if "c2w_nearest" in tensors and "w2c_nearest" not in tensors:
            self.inputs["w2c_nearest"].copy_(torch.inverse(self.inputs["c2w_nearest"].reshape(-1, 4, 4)))
        if frame_weight is not None:
            self.inputs["frame_weight"].fill_(float(frame_weight))
        self.graph.replay()
        if assign_grads:
            emb, conf, pdir, color = self.leaves
            _assign_grads(self.agg, emb, conf, pdir, color, self.pg, self.ag, accumulate=False)
        return self.out, self.pg, self.ag
