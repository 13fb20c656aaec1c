"""Blur-handling module on the device (SURVEY 8f "next" row 2): pre-defined kernels (blur_update_output) and learnable
per-patch kernels (learnable_blur_update_output).

Mirror of BaseRenderingModel.blur_update_output (/root/reference/models/base_rendering_model.py:677-745, faster_version),
which the training shell calls between the render and the losses when `add_blur_sim=1` and no learnable blur predictor is
active (mvs_points_volumetric_model.py:145-146).  One HIP block per patch replaces the two F.conv2d calls, the 5-D
tile/abs/sum, the fancy indexing and the Python re-assembly loop; the backward goes through the selected kernel only.
"""
import torch

from . import _lib
from ._lib import HnrError


def blur_select(color, gt, kernels, patch_num, patch_size):
    """hnr_blur_select without an autograd graph: (new colours [R,3], selected kernel per patch int32 [n_patches]).  patch_num < 0: -patch_num whole
    patches packed patch-major.  What train.train_step queues between the forward call and the loss kernels."""
    L = _lib.lib()
    c = _lib.require_gpu(color.detach(), "coarse_raycolor", torch.float32).reshape(-1, 3)
    g = _lib.require_gpu(gt, "gt_image", torch.float32).reshape(-1, 3)
    k = _lib.require_gpu(kernels, "blur_kernels", torch.float32)
    k = k.reshape(-1, k.shape[-2], k.shape[-1])
    if k.shape[-1] != k.shape[-2]:
        raise HnrError("blur kernels must be square")
    n_patches = -patch_num if patch_num < 0 else patch_num * patch_num          # patch_num < 0: patch-major list of whole patches
    if c.shape[0] != n_patches * patch_size * patch_size or g.shape[0] != c.shape[0]:
        raise HnrError("blur_update_output: expected %d patches of %dx%d rays, got %d rays" % (n_patches, patch_size, patch_size, c.shape[0]))
    out = torch.empty_like(c)
    sel = torch.empty((n_patches,), dtype=torch.int32, device=c.device)
    with torch.cuda.device(c.device):
        _lib.check(L.hnr_blur_select(_lib.ptr(c), _lib.ptr(g), _lib.ptr(k), k.shape[0], k.shape[-1], patch_num, patch_size, _lib.ptr(out),
                                     _lib.ptr(sel), _lib.stream()), "hnr_blur_select")
    return out, sel, k


def blur_select_bwd(g_out, k, sel, patch_num, patch_size):
    """d loss / d rendered colours through the selected kernels only (the selection itself is piecewise constant)."""
    L = _lib.lib()
    g = _lib.require_gpu(g_out.contiguous(), "grad", torch.float32).reshape(-1, 3)
    g_in = torch.empty_like(g)
    with torch.cuda.device(g.device):
        _lib.check(L.hnr_blur_select_bwd(_lib.ptr(g), _lib.ptr(k), _lib.ptr(sel), k.shape[0], k.shape[-1], patch_num, patch_size, _lib.ptr(g_in),
                                         _lib.stream()), "hnr_blur_select_bwd")
    return g_in


class _BlurSelect(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, gt, kernels, patch_num, patch_size):
        out, sel, k = blur_select(color, gt, kernels, patch_num, patch_size)
        ctx.save_for_backward(k, sel)
        ctx.dims = (patch_num, patch_size, color.shape)
        ctx.mark_non_differentiable(sel)
        return out.reshape(color.shape), sel

    @staticmethod
    def backward(ctx, g_out, _g_sel):
        k, sel = ctx.saved_tensors
        pn, ps, shape = ctx.dims
        return blur_select_bwd(g_out, k, sel, pn, ps).reshape(shape), None, None, None, None


def blur_update_output(coarse_raycolor, gt_image, blur_kernels, patch_num, patch_size, return_select=False, layout="grid"):
    """coarse_raycolor, gt_image: [1, S*S, 3] (S = patch_num * patch_size, dilated-patch ray layout); blur_kernels [1, N, ks, ks]
    (the dataset item's `blur_kernels`, data/scannet_ft_dataset.py:974).  Returns the new coarse_raycolor (same shape,
    differentiable w.r.t. the input colours).  layout="patch_major": the tensors hold `patch_num` whole patches packed
    (patch, y, x) -- a rank's share when the batch is sharded by patches (parallel.shard_patches)."""
    if layout not in ("grid", "patch_major"):
        raise HnrError("blur_update_output: layout must be 'grid' or 'patch_major'")
    out, sel = _BlurSelect.apply(coarse_raycolor, gt_image, blur_kernels, -int(patch_num) if layout == "patch_major" else int(patch_num),
                                 int(patch_size))
    return (out, sel) if return_select else out


# ----------------------------------------------------------------------------------------------------------------------
# Learnable blur kernels: BaseRenderingModel.learnable_blur_update_output (models/base_rendering_model.py:827-1020,
# faster_version), called at mvs_points_volumetric_model.py:146-147 when the aggregator hands out a blur predictor.
# ----------------------------------------------------------------------------------------------------------------------
def _pn(patch_num, layout):
    if layout not in ("grid", "patch_major"):
        raise HnrError("layout must be 'grid' or 'patch_major'")
    return -int(patch_num) if layout == "patch_major" else int(patch_num)


class _GrayPatches(torch.autograd.Function):
    """[n_patches, 2, ps, ps]: grey ground-truth patch, grey rendered patch (:886-893)."""

    @staticmethod
    def forward(ctx, color, gt, patch_num, patch_size):
        L = _lib.lib()
        c = _lib.require_gpu(color.detach(), "coarse_raycolor", torch.float32).reshape(-1, 3)
        g = _lib.require_gpu(gt, "gt_image", torch.float32).reshape(-1, 3)
        n = -patch_num if patch_num < 0 else patch_num * patch_num
        if c.shape[0] != n * patch_size * patch_size or g.shape[0] != c.shape[0]:
            raise HnrError("learnable_blur_update_output: expected %d patches of %dx%d rays, got %d rays" % (n, patch_size, patch_size, c.shape[0]))
        out = torch.empty((n, 2, patch_size, patch_size), dtype=torch.float32, device=c.device)
        with torch.cuda.device(c.device):
            _lib.check(L.hnr_blur_gray_patches(_lib.ptr(c), _lib.ptr(g), patch_num, patch_size, _lib.ptr(out), _lib.stream()), "hnr_blur_gray_patches")
        ctx.dims = (patch_num, patch_size, color.shape)
        return out

    @staticmethod
    def backward(ctx, g_gray):
        L = _lib.lib()
        pn, ps, shape = ctx.dims
        g = _lib.require_gpu(g_gray.contiguous(), "grad", torch.float32)
        g_c = torch.empty((g.shape[0] * ps * ps, 3), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _lib.check(L.hnr_blur_gray_patches_bwd(_lib.ptr(g), pn, ps, _lib.ptr(g_c), _lib.stream()), "hnr_blur_gray_patches_bwd")
        return g_c.reshape(shape), None, None, None


class _BlurApply(torch.autograd.Function):
    """Every patch convolved with its own kernel under opt.boundary_mode (:915-923); differentiable w.r.t. colours and kernels."""

    @staticmethod
    def forward(ctx, color, kernels, patch_num, patch_size, boundary_mode):
        L = _lib.lib()
        c = _lib.require_gpu(color.detach(), "coarse_raycolor", torch.float32).reshape(-1, 3)
        k = _lib.require_gpu(kernels.detach(), "blur_kernels", torch.float32)
        n = -patch_num if patch_num < 0 else patch_num * patch_num
        k = k.reshape(n, k.shape[-2], k.shape[-1])
        if k.shape[-1] != k.shape[-2] or c.shape[0] != n * patch_size * patch_size:
            raise HnrError("blur_apply: expected %d square kernels and %d rays" % (n, n * patch_size * patch_size))
        out = torch.empty_like(c)
        with torch.cuda.device(c.device):
            _lib.check(L.hnr_blur_apply(_lib.ptr(c), _lib.ptr(k), k.shape[-1], patch_num, patch_size, boundary_mode, _lib.ptr(out), _lib.stream()),
                       "hnr_blur_apply")
        ctx.save_for_backward(c, k)
        ctx.dims = (patch_num, patch_size, boundary_mode, color.shape, kernels.shape)
        return out.reshape(color.shape)

    @staticmethod
    def backward(ctx, g_out):
        L = _lib.lib()
        c, k = ctx.saved_tensors
        pn, ps, mode, cshape, kshape = ctx.dims
        g = _lib.require_gpu(g_out.contiguous(), "grad", torch.float32).reshape(-1, 3)
        g_c, g_k = torch.empty_like(c), torch.empty_like(k)
        with torch.cuda.device(g.device):
            _lib.check(L.hnr_blur_apply_bwd(_lib.ptr(g), _lib.ptr(c), _lib.ptr(k), k.shape[-1], pn, ps, mode, _lib.ptr(g_c), _lib.ptr(g_k),
                                            _lib.stream()), "hnr_blur_apply_bwd")
        return g_c.reshape(cshape), g_k.reshape(kshape), None, None, None


def learnable_blur_update_output(coarse_raycolor, gt_image, blur_predictor, opt, patch_num, patch_size, layout="grid", return_kernels=False):
    """coarse_raycolor, gt_image [1, S*S, 3] in the dilated-patch ray layout; blur_predictor = what the aggregator returns
    (PointAggregator.blur_predictor(): the MLP, or [conv block, MLP] with opt.learnable_blur_kernel_conv); opt fields read:
    learnable_blur_kernel_size, learnable_blur_kernel_norm, learnable_blur_kernel_mode (0 / 4), boundary_mode (0 / 1 / 2),
    learnable_blur_kernel_conv.  Returns the new coarse_raycolor, differentiable w.r.t. the colours and the predictor.
    The predictor (49 rows) and the kernel normalisation / identity blend (:897-911) run as torch ops on [N, ks*ks]; the
    patch gathering, the grouped convolution with its border rule and both backward passes are HIP kernels."""
    pn = _pn(patch_num, layout)
    ps, ks = int(patch_size), int(opt.learnable_blur_kernel_size)
    n = -pn if pn < 0 else pn * pn
    gray = _GrayPatches.apply(coarse_raycolor, gt_image, pn, ps)                       # [N, 2, ps, ps]
    if getattr(opt, "learnable_blur_kernel_conv", 0):
        pred = blur_predictor[1](blur_predictor[0](gray).view(n, -1))                  # :889
    else:
        pred = blur_predictor(gray.view(n, -1))                                        # :893 ([gt grey | render grey] per patch)
    if getattr(opt, "learnable_blur_kernel_norm", 0) == 0:                             # :897-901
        k = pred[:, 0:ks * ks].view(n, 1, ks, ks)
        k = k / torch.sum(k, dim=(2, 3), keepdim=True)
    else:
        k = torch.nn.functional.softmax(pred[:, 0:ks * ks], dim=-1).view(n, 1, ks, ks)
    mode = int(getattr(opt, "learnable_blur_kernel_mode", 0))
    if mode == 4:                                                                      # :906-910
        w = pred[:, -1][..., None, None, None]
        ident = torch.zeros_like(k)
        ident[:, :, ks // 2, ks // 2] = 1.0
        k = w * k + (1 - w) * ident
        k = k / torch.sum(k, dim=(2, 3), keepdim=True)
    elif mode != 0:
        raise HnrError("learnable_blur_kernel_mode %d is not implemented by the reference either (:911-912)" % mode)
    bm = int(getattr(opt, "boundary_mode", 0))
    if bm not in (0, 1, 2):
        raise HnrError("boundary_mode %d is not implemented by the reference either (:924-932)" % bm)
    out = _BlurApply.apply(coarse_raycolor, k, pn, ps, bm)
    return (out, k) if return_kernels else out
