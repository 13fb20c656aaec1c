"""torch.ops.hnr.*: the path as dispatcher-visible PyTorch ops (csrc/torch_ops/hnr_torch.cpp, TORCH_LIBRARY(hnr) over the C ABI of libhnr_hip.so).

SURVEY 8b asks for both op layers: the C ABI (include/hnr.h, bound by ctypes in _lib.py -- needs no torch headers) and registered torch ops.  The
ops are the same library calls with schemas: `hnr::grid_build`, `hnr::grid_free`, `hnr::march_query`, `hnr::render_forward`
(NeuralPointsRayMarching.forward + fill_invalid in eval mode, /root/reference/models/neural_points_volumetric_model.py:257-391, :87-126) and
`hnr::render_train` (the same in train mode with the backward pass registered as its autograd formula; the reference leaves that to torch autograd,
models/mvs_points_volumetric_model.py:111-131).  This module loads the extension and builds the ops' argument lists from the host-side objects
(HybridRenderer, PointAggregator); the results are bit-identical to the ctypes route (tests/test_torch_ops_gpu.py).  No fallback: a missing
libhnr_torch.so raises."""
import os

import numpy as np
import torch

from ._lib import HnrError
from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhnr_torch.so")
_loaded = False

# hnr_train_weights in struct order (include/hnr.h): parameter names of the reference's PointAggregator
TRAIN_WEIGHT_NAMES = (
    ["block1.0.weight", "block1.0.bias", "block1.2.weight", "block1.2.bias", "block3.0.weight", "block3.0.bias", "block3.2.weight", "block3.2.bias",
     "alpha_branch.0.weight", "alpha_branch.0.bias"]
    + ["color_feature_branch.%d.weight" % l for l in (0, 2, 4)] + ["color_feature_branch.%d.bias" % l for l in (0, 2, 4)]
    + ["aux_merge_weight_block.%d.weight" % l for l in (0, 2, 4, 6)] + ["aux_merge_weight_block.%d.bias" % l for l in (0, 2, 4, 6)]
    + ["color_mixup_block.%d.weight" % l for l in (0, 2, 4)] + ["color_mixup_block.%d.bias" % l for l in (0, 2, 4)]
    + ["color_final_block.0.weight", "color_final_block.0.bias"]
    + ["aux_block_s%d.%d.weight" % (s, l) for s in (1, 2, 3) for l in (0, 2)] + ["aux_block_s%d.%d.bias" % (s, l) for s in (1, 2, 3) for l in (0, 2)])
FORWARD_OUTPUTS = ("coarse_raycolor", "coarse_point_opacity", "coarse_is_background", "ray_mask", "decoded", "sample_pidx", "sample_loc_w", "ray_nsamp",
                   "counts", "status")
TRAIN_OUTPUTS = ("coarse_raycolor", "coarse_point_opacity", "coarse_is_background", "blend_weight", "ray_mask", "decoded", "sample_pidx", "sample_loc_w",
                 "ray_nsamp", "counts", "status", "weight", "conf_coefficient")


NCOUNTS = _lib.NCOUNTS


def _register_fakes():
    """Shape functions ("fake" / meta kernels) of the tensor-returning ops: what torch.compile, torch.export and FakeTensorMode need to trace through
    them without running a kernel.  Output shapes and dtypes are those of the C++ implementations (hnr_torch.cpp)."""
    f32, i32 = torch.float32, torch.int32

    @torch.library.register_fake("hnr::march_query")
    def _(grid, campos, raydir, tmid, SR, K, radius2, kernel_size, pad, knn_order):
        R, e = raydir.shape[0], raydir.new_empty
        return (e((R, SR, K), dtype=i32), e((R, SR, 3), dtype=f32), e((R,), dtype=i32), e((R,), dtype=torch.int8), e((NCOUNTS,), dtype=torch.int64))

    @torch.library.register_fake("hnr::render_forward")
    def _(grid, xyz, conf, dir, color, point_table, point_records, packed, campos, camrot, raydir, tmid, bg_color, w2c, intrinsic, campos_nearest, featmap,
          frame_w, SR, K, kernel_size, radius2, vsize_z, raydist_mode_unit, knn_order, slope, cap_samples):
        R, e = raydir.shape[0], raydir.new_empty
        return [e((R, 3), dtype=f32), e((R, SR), dtype=f32), e((R,), dtype=f32), e((R,), dtype=torch.int8), e((R, SR, 4), dtype=f32), e((R, SR, K), dtype=i32),
                e((R, SR, 3), dtype=f32), e((R,), dtype=i32), e((NCOUNTS,), dtype=torch.int64), e((2,), dtype=i32)]

    def train_outs(inputs, SR):
        raydir = inputs[7]
        R, K, e = raydir.shape[0], 8, raydir.new_empty
        return [e((R, 3), dtype=f32), e((R, SR), dtype=f32), e((R,), dtype=f32), e((R, SR), dtype=f32), e((R,), dtype=torch.int8), e((R, SR, 4), dtype=f32),
                e((R, SR, K), dtype=i32), e((R, SR, 3), dtype=f32), e((R,), dtype=i32), e((NCOUNTS,), dtype=torch.int64), e((2,), dtype=i32),
                e((R, SR, K), dtype=f32), e((R, SR, K), dtype=f32)]

    @torch.library.register_fake("hnr::render_train")
    def _(grid, inputs, weights, drop_lut, ray_drop, SR, kernel_size, radius2, vsize_z, raydist_mode_unit, knn_order, slope, cap_samples):
        return train_outs(inputs, SR)

    @torch.library.register_fake("hnr::render_train_fwd")
    def _(grid, inputs, weights, drop_lut, ray_drop, SR, kernel_size, radius2, vsize_z, raydist_mode_unit, knn_order, slope, cap_samples):
        import ctypes
        raydir, tmid, img = inputs[7], inputs[8], inputs[13]
        prm = _lib.TrainParams()                                # the workspace size is a host-side function of the shapes (no kernel runs)
        prm.R, prm.SR, prm.K, prm.D = int(raydir.shape[0]), int(SR), 8, int(tmid.shape[-1])
        prm.tmid_stride = 0 if tmid.dim() == 1 else int(tmid.shape[1])
        if img is not None and img.numel() > 0:
            prm.V, prm.H, prm.W = int(img.shape[0]), int(img.shape[1]), int(img.shape[2])
        for i in range(3):
            prm.kernel_size[i] = int(kernel_size[i])
        prm.radius2, prm.vsize_z, prm.raydist_mode_unit = float(radius2), float(vsize_z), int(raydist_mode_unit)
        prm.knn_order, prm.slope = int(knn_order), float(slope)
        prm.n_points = int(inputs[0].shape[-2])
        prm.cap_samples = int(cap_samples) if cap_samples > 0 else prm.R * prm.SR
        nbytes = int(_lib.lib().hnr_render_train_workspace_bytes(ctypes.byref(prm)))
        if nbytes < 0:
            raise HnrError("hnr_render_train_workspace_bytes: %s" % _lib.lib().hnr_last_error().decode("utf-8", "replace"))
        return train_outs(inputs, SR) + [raydir.new_empty((nbytes + 256,), dtype=torch.uint8)]

    @torch.library.register_fake("hnr::render_train_bwd")
    def _(inputs, weights, fwd, g_raycolor, g_conf_coefficient, SR, kernel_size, radius2, vsize_z, raydist_mode_unit, knn_order, slope, cap_samples):
        N, e, img = int(inputs[0].shape[-2]), inputs[7].new_empty, inputs[13]
        views = img is not None and img.numel() > 0
        image_branch = lambda i: (16 <= i < 24) or i >= 32
        return [e((N, 32), dtype=f32), e((N,), dtype=f32), e((N, 3), dtype=f32), e((N, 3), dtype=f32)] + \
               [torch.empty_like(w) if (views or not image_branch(i)) else w.new_empty((0,)) for i, w in enumerate(weights)]


def load():
    """Registers torch.ops.hnr (once), with shape functions for tracing.  libhnr_torch.so links libhnr_hip.so next to it."""
    global _loaded
    if not _loaded:
        if not os.path.exists(LIB_PATH):
            raise HnrError("%s is missing: build it with `make -C hybridneuralrendering_amd/csrc` (__graft_entry__.build())" % LIB_PATH)
        _lib.lib()                                         # the ctypes handle first: both bindings share ONE loaded libhnr_hip.so
        torch.ops.load_library(LIB_PATH)
        _register_fakes()
        _loaded = True
    return torch.ops.hnr


def _handle(grid):
    h = grid.handle
    return int(h.value if hasattr(h, "value") else h)


def train_weight_list(aggregator):
    """The 44 parameter tensors hnr::render_train takes, in hnr_train_weights order."""
    prm = dict(aggregator.named_parameters())
    return [prm[n] for n in TRAIN_WEIGHT_NAMES]


def render_forward(renderer, cloud, raydir, campos, camrot, bg_color, near, far, c2w_nearest, campos_nearest, intrinsic_nearest, images_nearest,
                   frame_weight=None, w2c_nearest=None):
    """HybridRenderer.render_rays through torch.ops.hnr.render_forward: same grid, packed weights, per-point table and feature map (all cached by the
    renderer), one registered op instead of the ctypes call.  Returns the output dict of render_rays."""
    ops = load()
    opt, agg = renderer.opt, renderer.agg
    g = _lib.require_gpu
    raydir = g(raydir, "raydir", torch.float32).reshape(-1, 3)
    campos, camrot, bg_color = g(campos, "campos", torch.float32).reshape(3), g(camrot, "camrotc2w", torch.float32).reshape(3, 3), g(bg_color, "bg_color", torch.float32).reshape(3)
    grid, hp = renderer.querier._grid_for(cloud.xyz[None])
    tmid = renderer.querier._tmid_for(float(near), float(far), opt.z_depth_dim, raydir.shape[0], raydir.device)
    fm = w2c = None
    if getattr(opt, "use_nearest", 4) != 0:
        fm = renderer.feature_map(images_nearest)
        c2w = g(c2w_nearest, "c2w_nearest", torch.float32).reshape(-1, 4, 4)
        w2c = (torch.inverse(c2w) if w2c_nearest is None else w2c_nearest).contiguous()
    pk, m3 = agg.packed(), agg.packed_mlp3()
    packed = [agg.packed_chain(), m3["cf"].packed, m3["mw"].packed, m3["mx"].packed, pk["mw_last_w"], pk["mw_last_b"], pk["fin_w"], pk["fin_b"]]
    packed = [t if t.dtype == torch.float32 else t.view(torch.uint8) for t in packed]
    out = ops.render_forward(_handle(grid), cloud.xyz, cloud.conf, cloud.dir, cloud.color, renderer.point_table(cloud), renderer.point_records(cloud), packed,
                             campos, camrot, raydir, tmid, bg_color, w2c,
                             None if fm is None else g(intrinsic_nearest, "intrinsic_nearest", torch.float32).reshape(3, 3),
                             None if fm is None else g(campos_nearest, "campos_nearest", torch.float32).reshape(-1, 3), fm,
                             None if frame_weight is None else g(frame_weight, "frame_weight", torch.float32).reshape(-1),
                             int(opt.SR), int(opt.K), [int(k) for k in opt.kernel_size], float(np.float32(hp[0] ** 2)), float(np.float32(opt.vsize[2])),
                             int(getattr(opt, "raydist_mode_unit", 0) > 0), 1 if renderer.knn_order == "sorted" else 0, float(pk["slope"]), 0)
    return dict(zip(FORWARD_OUTPUTS, out))


def render_train(renderer, aggregator, xyz, emb, conf, pdir, color, raydir, campos, camrot, bg_color, near, far, c2w_nearest, campos_nearest,
                 intrinsic_nearest, images_nearest, frame_weight_nearest=None, tmid=None, drop_lut=None, ray_drop=None):
    """Differentiable render of one ray batch through torch.ops.hnr.render_train (autograd formula registered in C++): emb / conf / pdir / color may be the
    reference's nn.Parameters ([1,N,32], [1,N,1], [1,N,3], [1,N,3]); the aggregator's parameters receive gradients.  Returns the 13 outputs as a dict;
    coarse_raycolor and conf_coefficient are attached to the autograd graph.  drop_lut: train.drop_lut(opt, R, device) for the patch-drop pattern."""
    ops = load()
    opt = renderer.opt
    g = _lib.require_gpu
    raydir = g(raydir, "raydir", torch.float32).reshape(-1, 3)
    R = raydir.shape[0]
    grid, hp = renderer.querier._grid_for(xyz.detach()[None] if xyz.dim() == 2 else xyz.detach())
    if tmid is None:
        tmid = renderer.querier._tmid_for(float(near), float(far), opt.z_depth_dim, R, raydir.device)
    views = getattr(opt, "use_nearest", 4) != 0
    img = w2c = intr = camn = None
    if views:
        img = g(images_nearest, "images_nearest", torch.float32)
        img = img[0] if img.dim() == 5 else img
        w2c = torch.inverse(g(c2w_nearest, "c2w_nearest", torch.float32).reshape(-1, 4, 4)).contiguous()
        intr, camn = g(intrinsic_nearest, "intrinsic_nearest", torch.float32).reshape(3, 3), g(campos_nearest, "campos_nearest", torch.float32).reshape(-1, 3)
    inputs = [xyz.reshape(-1, 3), emb, conf, pdir, color, g(campos, "campos", torch.float32).reshape(3), g(camrot, "camrotc2w", torch.float32).reshape(3, 3), raydir,
              g(tmid, "tmid", torch.float32), g(bg_color, "bg_color", torch.float32).reshape(3), w2c, intr, camn, img,
              None if frame_weight_nearest is None else g(frame_weight_nearest, "frame_weight_nearest", torch.float32).reshape(-1)]
    out = ops.render_train(_handle(grid), inputs, train_weight_list(aggregator), drop_lut, ray_drop, int(opt.SR), [int(k) for k in opt.kernel_size],
                           float(np.float32(hp[0] ** 2)), float(np.float32(opt.vsize[2])), int(getattr(opt, "raydist_mode_unit", 0) > 0), 0,
                           float(aggregator.block1[1].negative_slope), 0)
    return dict(zip(TRAIN_OUTPUTS, out))
