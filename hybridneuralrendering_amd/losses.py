"""The loss terms of the shipped training configurations on the device (SURVEY 8f "next" row 2, second half).

Mirror of the enabled part of BaseRenderingModel.compute_losses (models/base_rendering_model.py:1060-1245; items from
dev_scripts/w_scannet_etf/scene241.sh:146-151): masked colour MSE on `coarse_raycolor` + zero-one regulariser on
`conf_coefficient`, value and both gradients from two small HIP kernels instead of masked_select copies and an autograd graph.
"""
import torch

from . import _lib
from ._lib import HnrError


def _loss_call(color, conf, gt, ray_mask, zero_epsilon, w_color, w_zero_one, frame_weight, conf_rows, want_grads=True):
    """One hnr_shipped_loss[_rows] launch pair: returns (out4, g_color, g_conf) -- all device tensors, nothing is read back."""
    L = _lib.lib()
    c = _lib.require_gpu(color.detach(), "coarse_raycolor", torch.float32).reshape(-1, 3)
    g = _lib.require_gpu(gt, "gt_image", torch.float32).reshape(-1, 3)
    m = _lib.require_gpu(ray_mask, "ray_mask").reshape(-1)
    if m.dtype != torch.int8:
        m = m.to(torch.int8)
    x = _lib.require_gpu(conf.detach(), "conf_coefficient", torch.float32).reshape(-1)
    if g.shape[0] != c.shape[0] or m.shape[0] != c.shape[0]:
        raise HnrError("shipped_loss: coarse_raycolor, gt_image and ray_mask disagree on the number of rays")
    R = int(c.shape[0])
    if conf_rows and (R == 0 or x.shape[0] % R):
        raise HnrError("shipped_loss: conf_rows needs conf_coefficient with one row per ray of the batch")
    dev = c.device
    out = torch.empty((4,), dtype=torch.float32, device=dev)
    g_c, g_x = (torch.empty_like(c), torch.empty_like(x)) if want_grads else (None, None)
    scratch = torch.empty((int(L.hnr_shipped_loss_scratch_bytes()),), dtype=torch.uint8, device=dev)
    p = _lib.ptr
    with torch.cuda.device(dev):
        if isinstance(frame_weight, torch.Tensor):
            # the item's scalar on the device (one float): read by the kernel, so a captured step replays with other values
            if not conf_rows:
                raise HnrError("shipped_loss: a device frame_weight needs conf_rows=True")
            fw = _lib.require_gpu(frame_weight, "frame_weight", torch.float32).reshape(-1)
            _lib.check(L.hnr_shipped_loss_rows_fw(p(c), p(g), p(m), R, p(x), int(x.shape[0] // R), float(zero_epsilon), float(w_color), float(w_zero_one),
                                                  p(fw), p(out), p(g_c) if want_grads else None, p(g_x) if want_grads else None, p(scratch),
                                                  _lib.stream()), "hnr_shipped_loss_rows_fw")
        elif conf_rows:
            _lib.check(L.hnr_shipped_loss_rows(p(c), p(g), p(m), R, p(x), int(x.shape[0] // R), float(zero_epsilon), float(w_color), float(w_zero_one),
                                               float(frame_weight), p(out), p(g_c) if want_grads else None, p(g_x) if want_grads else None, p(scratch),
                                               _lib.stream()), "hnr_shipped_loss_rows")
        else:
            _lib.check(L.hnr_shipped_loss(p(c), p(g), p(m), R, p(x), x.shape[0], float(zero_epsilon), float(w_color), float(w_zero_one),
                                          float(frame_weight), p(out), p(g_c) if want_grads else None, p(g_x) if want_grads else None, p(scratch),
                                          _lib.stream()), "hnr_shipped_loss")
    return out, g_c, g_x


class _ShippedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, conf, gt, ray_mask, zero_epsilon, w_color, w_zero_one, frame_weight, conf_rows):
        out, g_c, g_x = _loss_call(color, conf, gt, ray_mask, zero_epsilon, w_color, w_zero_one, frame_weight, conf_rows)
        ctx.save_for_backward(g_c, g_x)
        ctx.shapes = (color.shape, conf.shape)
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g_total, _g_out):
        g_c, g_x = ctx.saved_tensors
        cs, xs = ctx.shapes
        return (g_c * g_total).reshape(cs), (g_x * g_total).reshape(xs), None, None, None, None, None, None, None


def _fw(frame_weight):
    """The dataset item's scalar loss weight (models/base_rendering_model.py:1205) as a Python float.  A device tensor is refused: reading it
    would be a host synchronisation inside the training step (convert it once where the item is loaded)."""
    if frame_weight is None:
        return 1.0
    if isinstance(frame_weight, torch.Tensor):
        if frame_weight.is_cuda:
            raise HnrError("frame_weight (the item's scalar loss weight) must be a Python float or a CPU tensor, not a device tensor: "
                           "the per-view weights of the feature merge are `frame_weight_nearest`")
        if frame_weight.numel() != 1:
            raise HnrError("frame_weight is the item's scalar loss weight (1 value), got %d values" % frame_weight.numel())
        return float(frame_weight.reshape(-1)[0])
    return float(frame_weight)


def shipped_loss(coarse_raycolor, conf_coefficient, gt_image, ray_mask, zero_epsilon, w_color=1.0, w_zero_one=1e-4, frame_weight=None, conf_rows=False):
    """Returns (loss_total, parts) with parts = tensor {total, colour MSE, zero-one mean, valid rays}; loss_total is differentiable
    w.r.t. coarse_raycolor [.., R, 3] and conf_coefficient (any shape).  frame_weight: the dataset item's scalar (or None).
    conf_rows: conf_coefficient holds one row per ray of the BATCH ([R, SR, K], what render_train returns) and the rows of rays with
    ray_mask = 0 are left out on the device -- instead of indexing it with the mask first (a masked copy, a host read of the number of
    valid rays, and an index_put in the backward pass)."""
    return _ShippedLoss.apply(coarse_raycolor, conf_coefficient, gt_image, ray_mask, zero_epsilon, w_color, w_zero_one, _fw(frame_weight), bool(conf_rows))


def shipped_loss_grads(coarse_raycolor, conf_coefficient, gt_image, ray_mask, zero_epsilon, w_color=1.0, w_zero_one=1e-4, frame_weight=None, conf_rows=True,
                       device_frame_weight=None):
    """The loss terms and their gradients without an autograd graph: (parts [4] = {total, colour MSE, zero-one mean, valid rays},
    d total / d coarse_raycolor [R,3], d total / d conf_coefficient (flat)) -- what train.train_step feeds to the backward pass.
    device_frame_weight: a one-float device tensor holding the item's scalar (a captured step; the kernel reads it)."""
    fw = device_frame_weight if device_frame_weight is not None else _fw(frame_weight)
    return _loss_call(coarse_raycolor, conf_coefficient, gt_image, ray_mask, zero_epsilon, w_color, w_zero_one, fw, bool(conf_rows))
