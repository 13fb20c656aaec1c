"""Whole-frame render driver: the counterpart of the per-chunk loop of the reference's test / eval drivers.

/root/reference/run/test_ft.py:146-209 (and run/train_ft.py:294-351) slices a frame's rays into `random_sample_size**2`
(= 2304) ray chunks, runs the model per chunk (grid rebuild, projection of all N points, 4-view CNN + 221 MB upsample, three
host syncs -- 124 times per 620x460 frame) and scatters every chunk into an [H,W,3] numpy image through `.cpu()`.
Here a frame is ONE launch of the fused path over all its rays (the grid and the reference-view pyramid are built once),
the [H,W,3] image is assembled on the device by pixel index, and with torch.distributed the rays are sharded over the ranks
and reassembled on rank 0 with one gather (parallel.render_sharded).  Chunking is still available (`chunk_rays`) and gives
the same pixels at jitter 0 (tests/test_render_gpu.py::test_whole_frame_equals_the_reference_chunk_loop, against a frame the imported
reference rendered chunk by chunk).
"""
import torch
import torch.distributed as dist

from ._lib import HnrError
from . import parallel


def render_image(renderer, cloud, frame, chunk_rays=0, sharded=None, group=None):
    """frame: dict with the dataset item of the reference (data/scannet_ft_dataset.py:855-976), batch dim optional:
    raydir [R,3], pixel_idx [R,2] (x, y), campos [3], camrotc2w [3,3], bg_color [3], near, far, h, w, c2w_nearest [V,4,4],
    campos_nearest [V,3], intrinsic_nearest [3,3], images_nearest [V,H,W,3] (+ optional frame_weight_nearest [V]).
    Returns dict(image [h,w,3] (zero where no ray was cast: run/test_ft.py:191 scatters into np.zeros), ray_mask [R] i8, coarse_raycolor [R,3]) --
    on rank 0 when sharded, None on the other ranks."""
    sq = lambda t, nd: t.reshape(t.shape[-nd:]) if isinstance(t, torch.Tensor) else t
    raydir = sq(frame["raydir"], 2)
    pix = sq(frame["pixel_idx"], 2)
    R = raydir.shape[0]
    if pix.shape[0] != R:
        raise HnrError("render_image: pixel_idx and raydir disagree on the number of rays")
    h, w = int(frame["h"]), int(frame["w"])
    near, far = float(torch.as_tensor(frame["near"]).min()), float(torch.as_tensor(frame["far"]).max())
    fw = frame.get("frame_weight_nearest")
    if sharded is None:
        sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1

    statuses = []

    def render(rays):
        outs = []
        step = rays.shape[0] if chunk_rays <= 0 else chunk_rays
        for lo in range(0, rays.shape[0], max(step, 1)):
            o = renderer.render_rays(cloud, rays[lo:lo + step].contiguous(), sq(frame["campos"], 1), sq(frame["camrotc2w"], 2),
                                     sq(frame["bg_color"], 1), near, far, sq(frame["c2w_nearest"], 3), sq(frame["campos_nearest"], 2),
                                     sq(frame["intrinsic_nearest"], 2), sq(frame["images_nearest"], 4),
                                     frame_weight=None if fw is None else sq(fw, 1))
            statuses.append(o)
            outs.append(torch.cat([o["coarse_raycolor"], o["ray_mask"].to(torch.float32)[:, None]], dim=1))
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)

    def render_auto(rays):
        """whole block in one launch; when the workspace does not fit (render_rays says so) halve the chunk until it does"""
        nonlocal chunk_rays
        for _ in range(8):
            try:
                return render(rays)
            except HnrError as e:
                if "in chunks" not in str(e):
                    raise
                chunk_rays = max((rays.shape[0] if chunk_rays <= 0 else chunk_rays) // 2, 1)
        return render(rays)

    rows = parallel.render_sharded(render_auto, raydir, group=group) if sharded else render_auto(raydir)
    # the device status words of this rank's launches, read once per frame (after everything is queued): an overflow of a workspace capacity
    # must not pass silently
    renderer.check_status([dict(status=o.get("status")) for o in statuses])
    if rows is None:
        return None
    col, mask = rows[:, :3].contiguous(), rows[:, 3].to(torch.int8)
    img = torch.zeros((h, w, 3), dtype=torch.float32, device=col.device)        # test_ft.py:191 `np.zeros((height, width, 3))`: the margin stays zero
    px, py = pix[:, 0].to(col.device, torch.long), pix[:, 1].to(col.device, torch.long)
    img[py, px] = col                                            # test_ft.py:193 `visuals[key][y, x, :] = chunk`, on the device
    return dict(image=img, ray_mask=mask, coarse_raycolor=col)
