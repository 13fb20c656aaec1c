"""Drop-in nn.Module surface of the reference's hot path, backed by libhnr_hip.so.

    NeuralPoints             <- models/neural_points/neural_points.py:11-733
    ray_march, find_*        <- models/rendering/diff_ray_marching.py:508-557, diff_render_func.py:8-63
    NeuralPointsRayMarching  <- models/neural_points_volumetric_model.py:219-427
    install()                rebinds the reference's module globals so that
                             run/train_ft.py / run/test_ft.py pick these classes up with no edits.

Same constructor arguments, parameter names, call signatures and return tuples / dict keys.  Forward only in
round 1: the outputs carry no autograd graph (training through these modules raises in `backward`-requiring
code paths because the tensors do not require grad).
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import HnrError
from . import querier as Q
from .aggregator import PointAggregator
from .render import HybridRenderer, PointCloud


# ------------------------------------------------------------------------------------------------ rendering
def alpha_blend(opacity, acc_transmission):
    return opacity * acc_transmission


def radiance_render(ray_feature):
    return ray_feature[..., 1:]


def no_tone_map(color, gamma=2.2, exposure=1):
    return color


def find_render_function(name):
    if name == "radiance":
        return radiance_render
    raise RuntimeError("Unknown / unsupported render function: " + name)


def find_blend_function(name):
    if name == "alpha":
        return alpha_blend
    raise RuntimeError("Unknown / unsupported blend function: " + name)


def find_tone_map(name):
    if name == "off":
        return no_tone_map
    raise RuntimeError("Unknown / unsupported tone map: " + name)


def ray_march(ray_dist, ray_valid, ray_features, render_func, blend_func, bg_color=None):
    """diff_ray_marching.py:508-557 for tensors that already exist (drop-in for callers that hold decoded features);
    runs hnr_ray_march.  The fused NeuralPointsRayMarching path composites inside hnr_composite and differentiates through
    hnr_composite_bwd; this stand-alone operator is forward-only."""
    if render_func is not radiance_render or blend_func is not alpha_blend:
        raise HnrError("ray_march: only radiance render / alpha blend are implemented (all shipped configs)")
    if torch.is_grad_enabled() and (ray_features.requires_grad or ray_dist.requires_grad):
        raise HnrError("ray_march: the stand-alone operator is forward-only; train through NeuralPointsRayMarching (HIP backward)")
    feats = _lib.require_gpu(ray_features, "ray_features", torch.float32)
    if feats.dim() != 4 or feats.shape[-1] != 4:
        raise HnrError("ray_march: ray_features must be [N, R, SR, 4] (sigma, rgb)")
    N, R, SR, _ = feats.shape
    dist = _lib.require_gpu(ray_dist.to(torch.float32), "ray_dist", torch.float32).reshape(N * R, SR)
    valid = _lib.require_gpu(ray_valid.to(torch.uint8), "ray_valid", torch.uint8).reshape(N * R, SR)
    dev = feats.device
    f = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
    out = []
    for n in range(N):                                           # N (batch) is 1 everywhere; bg_color is per batch entry
        col, opa, acc, bw, bgt = f(R, 3), f(R, SR), f(R, SR), f(R, SR), f(R)
        bg = None if bg_color is None else _lib.require_gpu(bg_color.to(dev).float().reshape(-1, 3)[n if bg_color.numel() > 3 else 0],
                                                            "bg_color", torch.float32)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().hnr_ray_march(_lib.ptr(dist[n * R:(n + 1) * R]), _lib.ptr(valid[n * R:(n + 1) * R]), _lib.ptr(feats[n]),
                                                _lib.ptr(bg) if bg is not None else None, R, SR, _lib.ptr(col), _lib.ptr(opa), _lib.ptr(acc),
                                                _lib.ptr(bw), _lib.ptr(bgt), _lib.stream()), "hnr_ray_march")
        out.append((col, opa, acc, bw, bgt))
    cat = lambda i: torch.stack([o[i] for o in out], dim=0)
    bg_t = cat(4)[..., None]
    return cat(0), feats[..., 1:], cat(1), cat(2), cat(3)[..., None], bg_t, bg_t


# ------------------------------------------------------------------------------------------------ NeuralPoints
class NeuralPoints(nn.Module):
    def __init__(self, num_channels, size, opt, device, checkpoint=None, feature_init_method="rand", reg_weight=0., feedforward=0):
        super().__init__()
        assert isinstance(size, int), "size must be int"
        self.opt = opt
        self.grid_vox_sz = 0
        self.points_conf, self.points_dir, self.points_color, self.eulers, self.Rw2c = None, None, None, None, None
        self.device = device
        if getattr(opt, "wcoord_query", 1) <= 0:
            raise HnrError("only the world-coordinate querier (wcoord_query=1) is implemented; no shipped script uses the other")
        if getattr(opt, "xyz_grad", 0) > 0:
            # the reference cannot run this either: with xyz requiring grad its query_points calls .cpu().numpy() on the
            # point bounds (query_point_indices_worldcoords.py:67, reached from neural_points.py:568) and raises a
            # RuntimeError on the first ray batch (reproduced with the imported reference).  Raise up front instead.
            raise HnrError("xyz_grad > 0 is not supported: the reference's world-coordinate querier raises on the first query with it "
                           "(numpy() on a tensor that requires grad), and the HIP backward carries no position gradient")
        fresh_conf = False
        if getattr(opt, "load_points", 0) == 1:
            saved = torch.load(checkpoint, map_location=device) if checkpoint else None
            if saved is None or "neural_points.xyz" not in saved:
                # no cloud in the checkpoint (:248-308): positions from opt.cloud_path, features from feature_init_method, confidence 1
                from . import cloud_io
                if getattr(opt, "construct_res", 0) > 0 or len(getattr(opt, "point_noise", "")) > 0:
                    raise HnrError("NeuralPoints: construct_res / point_noise are not implemented (no shipped script sets them)")
                xyz = torch.as_tensor(cloud_io.load_cloud(opt), device=device, dtype=torch.float32)
                emb, conf = cloud_io.init_point_features(xyz, num_channels, feature_init_method, device, int(opt.point_features_dim))
                saved = dict(saved or {})
                saved.update({"neural_points.xyz": xyz})
                saved.setdefault("neural_points.points_embeding", emb)
                if "neural_points.points_conf" not in saved and not checkpoint:
                    saved["neural_points.points_conf"] = conf
                    fresh_conf = True
            par = lambda k, g: nn.Parameter(saved[k], requires_grad=g) if k in saved else None
            self.xyz = nn.Parameter(saved["neural_points.xyz"], requires_grad=opt.xyz_grad > 0)
            self.points_embeding = par("neural_points.points_embeding", opt.feat_grad > 0)
            if fresh_conf:
                # a cloud initialised from a file: the reference keeps points_conf as a plain ones tensor (neural_points.py:303) -- not a Parameter,
                # not in the state_dict, never seen by an optimiser
                self.points_conf = saved["neural_points.points_conf"]
            else:
                self.points_conf = par("neural_points.points_conf", opt.conf_grad > 0)
            self.points_dir = par("neural_points.points_dir", opt.dir_grad > 0)
            self.points_color = par("neural_points.points_color", opt.color_grad > 0)
            self.Rw2c = torch.eye(3, device=self.xyz.device, dtype=self.xyz.dtype)
            if "neural_points.Rw2c" in saved or "neural_points.eulers" in saved:
                raise HnrError("per-point rotations (Rw2c / eulers) are never produced by a shipped config and are unsupported")
        self.reg_weight = reg_weight
        self.opt.query_size = self.opt.kernel_size if self.opt.query_size[0] == 0 else self.opt.query_size   # :329
        self.lighting_fast_querier = Q.lighting_fast_querier
        self.querier = self.lighting_fast_querier(device, self.opt)

    def reset_querier(self):
        self.querier.clean_up()
        del self.querier
        self.querier = self.lighting_fast_querier(self.device, self.opt)

    def _set(self, name, tensor, grad_flag):
        setattr(self, name, None if tensor is None else nn.Parameter(tensor, requires_grad=grad_flag > 0))
        if name == "xyz":
            self._points_changed()

    def _points_changed(self):
        """The voxel grid is cached per cloud (the reference rebuilds it per chunk, so it never has this problem): a change of
        the point set other than appending (prune, set_points: point ids shift) drops it and the next query rebuilds it;
        grow_points extends it in place (querier.grow)."""
        if getattr(self, "querier", None) is not None:
            self.querier.clean_up()

    def prune(self, thresh):                                           # :350-373
        mask = self.points_conf[0, ..., 0] >= thresh
        self._set("xyz", self.xyz[mask, :], self.opt.xyz_grad)
        for name, flag in (("points_embeding", self.opt.feat_grad), ("points_conf", self.opt.conf_grad),
                           ("points_dir", self.opt.dir_grad), ("points_color", self.opt.color_grad)):
            t = getattr(self, name)
            if t is not None:
                self._set(name, t[:, mask, :], flag)
        print("@@@@@@@@@  pruned {}/{}".format(torch.sum(mask == 0), mask.shape[0]))

    def grow_points(self, add_xyz, add_embedding, add_color, add_dir, add_conf, add_eulers=None, add_Rw2c=None):   # :376-402
        n_old = int(self.xyz.shape[0])
        self.xyz = nn.Parameter(torch.cat([self.xyz, add_xyz], dim=0), requires_grad=self.opt.xyz_grad > 0)
        # points were APPENDED: the cached voxel grid is extended in place when the grown cloud keeps the grid geometry (SURVEY 8f-3;
        # querier.grow -> hnr_grid_grow), dropped and rebuilt by the next query otherwise
        if getattr(self, "querier", None) is not None:
            self.querier.grow(self.xyz, n_old)
        for name, add, flag in (("points_embeding", add_embedding, self.opt.feat_grad), ("points_conf", add_conf, self.opt.conf_grad),
                                ("points_dir", add_dir, self.opt.dir_grad), ("points_color", add_color, self.opt.color_grad)):
            t = getattr(self, name)
            if t is not None:
                self._set(name, torch.cat([t, add[None, ...]], dim=1), flag)

    def set_points(self, points_xyz, points_embeding, points_color=None, points_dir=None, points_conf=None, parameter=False,
                   Rw2c=None, eulers=None):                            # :404-470, modes "1" only
        if Rw2c is not None:
            raise HnrError("per-point Rw2c is unsupported")
        if points_embeding.shape[-1] > self.opt.point_features_dim:
            points_embeding = points_embeding[..., :self.opt.point_features_dim]
        if self.opt.default_conf > 0.0 and self.opt.default_conf <= 1.0 and points_conf is not None:
            points_conf = torch.ones_like(points_conf) * self.opt.default_conf
        wrap = (lambda t, f: nn.Parameter(t, requires_grad=f > 0)) if parameter else (lambda t, f: t)
        self.xyz = wrap(points_xyz, self.opt.xyz_grad)
        self.points_conf = None if points_conf is None else wrap(points_conf, self.opt.conf_grad)
        self.points_dir = None if points_dir is None else wrap(points_dir, self.opt.dir_grad)
        self.points_color = None if points_color is None else wrap(points_color, self.opt.color_grad)
        self.points_embeding = wrap(points_embeding, self.opt.feat_grad)
        self.Rw2c = torch.eye(3, device=points_xyz.device, dtype=points_xyz.dtype)
        self._points_changed()

    def editing_set_points(self, points_xyz, points_embeding, points_color=None, points_dir=None, points_conf=None,
                           parameter=False, Rw2c=None, eulers=None):   # :473-487
        if self.opt.default_conf > 0.0 and self.opt.default_conf <= 1.0 and points_conf is not None:
            points_conf = torch.ones_like(points_conf) * self.opt.default_conf
        self.xyz, self.points_embeding, self.points_dir = points_xyz, points_embeding, points_dir
        self.points_conf, self.points_color = points_conf, points_color
        self.Rw2c = torch.eye(3, device=points_xyz.device, dtype=points_xyz.dtype)
        self._points_changed()

    def null_grad(self):
        self.points_embeding.grad = None
        self.xyz.grad = None

    def reg_loss(self):
        return self.reg_weight * torch.mean(torch.pow(self.points_embeding, 2))

    def cloud(self):
        return PointCloud(self.xyz, self.points_embeding, self.points_conf, self.points_dir, self.points_color)

    def w2pers(self, point_xyz, camrotc2w, campos):                    # :607-613 (kept; the kernels fuse it)
        shift = point_xyz[None, ...] - campos[:, None, :]
        xyz = torch.sum(camrotc2w[:, None, :, :] * shift[:, :, :, None], dim=-2)
        return torch.stack([xyz[:, :, 0] / xyz[:, :, 2], xyz[:, :, 1] / xyz[:, :, 2], xyz[:, :, 2]], dim=-1)

    def forward(self, inputs):
        """:702-733 -> the 14-tuple (sampled_color, sampled_Rw2c, sampled_dir, sampled_conf, sampled_embedding,
        sampled_xyz_pers, sampled_xyz, sample_pnt_mask, sample_loc, sample_loc_w, sample_ray_dirs, ray_mask, vsize, grid_vox_sz)."""
        L = _lib.lib()
        camrot, campos = inputs["camrotc2w"], inputs["campos"]
        near, far = torch.min(inputs["near"]).item(), torch.max(inputs["far"]).item()
        pidx, loc, loc_w, dirs, ray_mask, vsize, _ = self.querier.query_points(
            inputs["pixel_idx"], None, self.xyz[None, ...], None, None, None, None, near, far, inputs["raydir"], campos, camrot)
        B, R, SR, K = pidx.shape
        n = B * R * SR * K
        dev = pidx.device
        c = self.cloud()
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        o_color, o_dir, o_conf, o_emb = f(B, R, SR, K, 3), f(B, R, SR, K, 3), f(B, R, SR, K, 1), f(B, R, SR, K, c.F)
        o_pers, o_xyz = f(B, R, SR, K, 3), f(B, R, SR, K, 3)
        o_mask = torch.empty((B, R, SR, K), dtype=torch.uint8, device=dev)
        p = _lib.ptr
        with torch.cuda.device(dev):
            _lib.check(L.hnr_gather_points(p(pidx), n, p(c.xyz), p(c.emb), p(c.conf), p(c.dir), p(c.color), c.F,
                                           p(campos.reshape(3).contiguous()), p(camrot.reshape(3, 3).contiguous()), p(o_color),
                                           p(o_dir), p(o_conf), p(o_emb), p(o_pers), p(o_xyz), p(o_mask), _lib.stream()),
                       "hnr_gather_points")
        return (o_color, self.Rw2c, o_dir, o_conf, o_emb, o_pers, o_xyz, o_mask.bool(), loc, loc_w, dirs, ray_mask, vsize,
                self.grid_vox_sz)


# ------------------------------------------------------------------------------------------------ ray marching
class NeuralPointsRayMarching(nn.Module):
    """Same constructor and forward signature / output dict as the reference (:219-391); the forward runs the fused
    HIP path (HybridRenderer) and then compacts to the reference's valid-ray row layout."""

    def __init__(self, tonemap_func=None, render_func=None, blend_func=None, aggregator=None, is_compute_depth=False,
                 neural_points=None, opt=None, num_pos_freqs=0, num_viewdir_freqs=0, **kwargs):
        super().__init__()
        self.aggregator = aggregator
        self.num_pos_freqs, self.num_viewdir_freqs = num_pos_freqs, num_viewdir_freqs
        self.render_func, self.blend_func, self.tone_map = render_func, blend_func, tonemap_func
        self.return_depth, self.return_color = is_compute_depth, True
        self.opt = opt
        self.neural_points = neural_points
        if is_compute_depth:
            raise HnrError("compute_depth is unsupported (no shipped config sets it; the reference path itself references an undefined ray_ts)")
        self._renderer = None
        self._train_path = None

    def renderer(self):
        if self._renderer is None:
            dev = self.neural_points.xyz.device
            self._renderer = HybridRenderer(self.opt, self.aggregator, dev)
            self._renderer.querier = self.neural_points.querier        # one grid cache for both surfaces
        return self._renderer

    def forward(self, campos, raydir, gt_image=None, bg_color=None, camrotc2w=None, pixel_idx=None, near=None, far=None,
                focal=None, h=None, w=None, intrinsic=None, aux_image=None, c2w=None, c2w_nearest=None, images_nearest=None,
                campos_nearest=None, intrinsic_nearest=None, vid_angle_nearest=None, frame_weight_nearest=None, **kargs):
        if "bg_ray" in kargs:
            raise HnrError("per-ray backgrounds (bg_ray) are unsupported")
        # kargs["tmid"] (optional, [R, z_depth_dim]): explicit marched depths instead of freshly drawn jitter (tests)
        rnd = self.renderer()
        nearv, farv = torch.min(near).item(), torch.max(far).item()
        fw = frame_weight_nearest[0] if getattr(self.opt, "downweight_blurry_feats", 0) else None
        if getattr(self.opt, "is_train", 0) and torch.is_grad_enabled():
            # training: same HIP forward with the activations kept + the hand-written backward (train.py); coarse_raycolor and
            # conf_coefficient stay attached to the autograd graph of the point buffers and the aggregator parameters
            from .train import TrainPath, render_train
            if self._train_path is None:
                self._train_path = TrainPath(rnd)
            npnt = self.neural_points
            full = render_train(self._train_path, self.aggregator, npnt.xyz, npnt.points_embeding, npnt.points_conf, npnt.points_dir,
                                npnt.points_color, raydir[0], campos[0], camrotc2w[0], bg_color[0], nearv, farv, c2w_nearest[0],
                                campos_nearest[0], intrinsic_nearest[0], images_nearest[0], frame_weight=fw, tmid=kargs.get("tmid"))
        else:
            cloud = self.neural_points.cloud()
            full = rnd.render_rays(cloud, raydir[0], campos[0], camrotc2w[0], bg_color[0], nearv, farv, c2w_nearest[0], campos_nearest[0],
                                   intrinsic_nearest[0], images_nearest[0], frame_weight=fw, want_weights=True, pad=True)
        mask = full["ray_mask"]
        rows = torch.nonzero(mask)[:, 0]                                  # valid rays, in ray order (:705-709)
        sel = lambda t: t.index_select(0, rows)[None]
        # the blur-kernel predictor modules go to the training shell untouched (:312-330: only in train mode)
        out = {"blur_predictor": self.aggregator.blur_predictor() if getattr(self.opt, "is_train", 0) else None}
        ray_valid = (full["sample_pidx"][..., 0] >= 0)
        out["queried_shading"] = torch.logical_not(torch.any(sel(ray_valid), dim=-1, keepdims=True)).repeat(1, 1, 3).to(torch.float32)
        col = sel(full["coarse_raycolor"])
        out["coarse_raycolor_patch"] = col
        out["coarse_raycolor"] = col
        out["coarse_point_opacity"] = sel(full["coarse_point_opacity"])
        out["coarse_is_background"] = sel(full["coarse_is_background"])[..., None]
        out["ray_mask"] = mask[None]
        out["weight"] = sel(full["weight"])
        out["blend_weight"] = sel(full["blend_weight"])[..., None]
        out["conf_coefficient"] = sel(full["conf_coefficient"])
        if getattr(self.opt, "prob", 0) == 1 and rows.numel() > 0:
            out.update(self._probe_outputs(full, rows))
        return out

    def _probe_outputs(self, full, rows):
        """opt.prob == 1 (:392-416): per valid ray the max-opacity sample and the weighted averages of its neighbours' attributes."""
        npnt = self.neural_points
        cloud = npnt.cloud()
        R, SR, K = full["sample_pidx"].shape
        dev = full["sample_pidx"].device
        f = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
        mo, loc, far, col, dr, cf, em = f(R), f(R, 3), f(R), f(R, 3), f(R, 3), f(R), f(R, cloud.F)
        p = _lib.ptr
        g = lambda t: _lib.require_gpu(t.detach(), "probe input", torch.float32)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().hnr_probe_outputs(p(g(full["coarse_point_opacity"])), p(full["sample_loc_w"]), p(full["sample_pidx"]), p(g(full["weight"])),
                                                    p(g(full["conf_coefficient"])), p(cloud.xyz), p(cloud.emb), p(cloud.conf), p(cloud.dir), p(cloud.color),
                                                    cloud.F, R, SR, K, p(mo), p(loc), p(far), p(col), p(dr), p(cf), p(em), _lib.stream()),
                       "hnr_probe_outputs")
        sel = lambda t: t.index_select(0, rows)[None]
        return dict(ray_max_shading_opacity=sel(mo)[..., None], ray_max_sample_loc_w=sel(loc), ray_max_far_dist=sel(far)[..., None],
                    shading_avg_color=sel(col), shading_avg_dir=sel(dr), shading_avg_conf=sel(cf)[..., None], shading_avg_embedding=sel(em))


def install():
    """Rebind the reference's module globals to these classes (call BEFORE create_model(opt)); needs the reference
    repository on sys.path.  See INTEGRATION.md."""
    import importlib
    vol = importlib.import_module("models.neural_points_volumetric_model")
    npm = importlib.import_module("models.neural_points.neural_points")
    vol.NeuralPoints = NeuralPoints
    vol.PointAggregator = PointAggregator
    vol.NeuralPointsRayMarching = NeuralPointsRayMarching
    vol.ray_march = ray_march
    npm.lighting_fast_querier_w = Q.lighting_fast_querier
    return vol
