"""Blur-handling module with pre-defined kernels on the device (SURVEY 8f "next" row 2).

Mirror of BaseRenderingModel.blur_update_output (/root/reference/models/base_rendering_model.py:677-745, faster_version),
which the training shell calls between the render and the losses when `add_blur_sim=1` and no learnable blur predictor is
active (mvs_points_volumetric_model.py:145-146).  One HIP block per patch replaces the two F.conv2d calls, the 5-D
tile/abs/sum, the fancy indexing and the Python re-assembly loop; the backward goes through the selected kernel only.
"""
import torch

from . import _lib
from ._lib import HnrError


class _BlurSelect(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, gt, kernels, patch_num, patch_size):
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
        ctx.save_for_backward(k, sel)
        ctx.dims = (patch_num, patch_size, color.shape)
        ctx.mark_non_differentiable(sel)
        return out.reshape(color.shape), sel

    @staticmethod
    def backward(ctx, g_out, _g_sel):
        L = _lib.lib()
        k, sel = ctx.saved_tensors
        pn, ps, shape = ctx.dims
        g = _lib.require_gpu(g_out.contiguous(), "grad", torch.float32).reshape(-1, 3)
        g_in = torch.empty_like(g)
        with torch.cuda.device(g.device):
            _lib.check(L.hnr_blur_select_bwd(_lib.ptr(g), _lib.ptr(k), _lib.ptr(sel), k.shape[0], k.shape[-1], pn, ps, _lib.ptr(g_in),
                                             _lib.stream()), "hnr_blur_select_bwd")
        return g_in.reshape(shape), None, None, None, None


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
