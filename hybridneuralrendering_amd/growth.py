"""Point-set maintenance on the device: hole probing -> new points.

Mirror of `probe_hole` in /root/reference/run/train_ft.py:450-569 without its driver plumbing (datasets, chunk loop, visualiser): the
caller renders the probed frames with `opt.prob = 1` (the ray_max_* / shading_avg_* outputs of NeuralPointsRayMarching,
models/neural_points_volumetric_model.py:392-416; here hnr_probe_outputs) and hands the per-ray outputs over; the selection of the
pixels that become points (:527-549: missed-ray neighbourhood, far-distance rule, opacity threshold) runs in csrc/probe.hip, the new
points' attributes are the selected rays' outputs (:551-560).  `NeuralPoints.grow_points` (modules.py) then appends them and drops the
cached voxel grid, so training continues in the same process -- the reference saves and exit()s here (:926-952).
"""
import ctypes

import torch

from . import _lib
from ._lib import HnrError


def probe_select(output, pixel_idx, gt_image, bg_color, height, width, far_thresh=0.0, opacity_thresh=0.7):
    """Indices (into the frame's rays) of the pixels that become new points, in the reference's order (row-major over the image).
    output: the prob == 1 output dict of one frame ([1, R, C] tensors; ray_mask [1, R]); pixel_idx [1, R, 2] (x, y); gt_image [R, 3]."""
    L, p = _lib.lib(), _lib.ptr
    g = lambda t, n: _lib.require_gpu(t.to(torch.float32), n, torch.float32)
    pix = g(pixel_idx, "pixel_idx").reshape(-1, 2)
    R = pix.shape[0]
    rm = g(output["ray_mask"], "ray_mask").reshape(-1)
    col = g(output["coarse_raycolor"], "coarse_raycolor").reshape(-1, 3)
    far = g(output["ray_max_far_dist"], "ray_max_far_dist").reshape(-1)
    opa = g(output["ray_max_shading_opacity"], "ray_max_shading_opacity").reshape(-1)
    gt = g(gt_image, "gt_image").reshape(-1, 3)
    if not (rm.shape[0] == col.shape[0] == far.shape[0] == opa.shape[0] == gt.shape[0] == R):
        raise HnrError("probe_select: per-ray tensors disagree on the number of rays")
    dev = pix.device
    bg = (ctypes.c_float * 3)(*[float(v) for v in torch.as_tensor(bg_color).reshape(-1)[:3].tolist()])
    miss = torch.empty((height * width,), dtype=torch.int32, device=dev)
    sel = torch.empty((height * width,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.hnr_probe_select(p(pix), p(rm), p(gt), p(col), bg, p(far), p(opa), R, int(height), int(width), float(far_thresh),
                                      float(opacity_thresh), p(miss), p(sel), _lib.stream()), "hnr_probe_select")
    return sel[sel > 0].long() - 1                    # boolean indexing walks the map in row-major order, like the reference's mask indexing (:551)


def probe_hole(frames, height, width, far_thresh=0.0, opacity_thresh=0.7, prob_mul=1.0):
    """frames: iterable of (output dict, pixel_idx [1,R,2], gt_image [R,3], bg_color [3]) in visiting order.
    Returns (add_xyz, add_embedding, add_color, add_dir, add_conf) as `probe_hole` does, including the reference's accumulation rule
    `add_conf = cat([add_conf, new]) * prob_mul` (:553-554), which rescales the earlier frames' confidences once more per later frame."""
    xyz = emb = col = dr = conf = None
    cat = lambda a, b: b if a is None else torch.cat([a, b], dim=0)
    for output, pixel_idx, gt_image, bg in frames:
        ids = probe_select(output, pixel_idx, gt_image, bg, height, width, far_thresh, opacity_thresh)
        take = lambda k: output[k].reshape(-1, output[k].shape[-1]).index_select(0, ids)
        xyz = cat(xyz, take("ray_max_sample_loc_w"))
        # a cloud without confidence / colour / direction buffers renders these outputs as None: the reference then returns None for them
        # (run/train_ft.py:545-550)
        conf = cat(conf, take("shading_avg_conf")) * prob_mul if output.get("shading_avg_conf") is not None else None
        col = cat(col, take("shading_avg_color")) if output.get("shading_avg_color") is not None else None
        dr = cat(dr, take("shading_avg_dir")) if output.get("shading_avg_dir") is not None else None
        emb = cat(emb, take("shading_avg_embedding"))
    if xyz is None:
        raise HnrError("probe_hole: no frame given")
    return xyz, emb, col, dr, conf
