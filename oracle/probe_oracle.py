"""CPU restatement of the hole-probing selection -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Follows /root/reference/run/train_ft.py:450-569 (`probe_hole`, the per-frame mask logic at :527-549 and the accumulation at
:551-560) and :571-581 (`bloat_inds`).  PINNED: tests/golden/probe_hole.npz holds outputs of the reference function itself
(tests/golden/make_golden.py::gen_probe_hole); tests/test_probe.py checks this restatement against it.
Only tests/ may import this."""
import numpy as np


def frame_selection(pix, ray_mask, raycolor, far_dist, opacity, gt, bg, h, w, far_thresh, opacity_thresh):
    """pix [R,2] (x, y) of the cast rays in row-major pixel order; per-ray arrays.  Returns the indices (into the rays) of the new
    points, in the reference's order (row-major over the [h, w] mask, :549-551)."""
    px, py = pix[:, 0].astype(np.int64), pix[:, 1].astype(np.int64)
    cast = np.zeros((h, w), bool); cast[py, px] = True                                   # edge_mask (:505-507)
    hit = np.zeros((h, w), np.float32); hit[py, px] = ray_mask
    gtm = np.zeros((h, w, 3), np.float32); gtm[py, px] = gt                             # :531-533
    miss = cast & (hit < 1) & (np.sqrt(((gtm - bg.reshape(1, 1, 3)) ** 2).sum(-1)) > 0.002)      # :534-535
    near = np.zeros((h, w), np.float32)
    ys, xs = np.nonzero(miss)
    for dy in (-1, 0, 1):                                                                # bloat_inds(shift = 1), clamped (:571-581)
        for dx in (-1, 0, 1):
            near[np.clip(ys + dy, 0, h - 1), np.clip(xs + dx, 0, w - 1)] = 1
    if far_thresh > 0:                                                                   # :540-543
        colm = np.zeros((h, w, 3), np.float32); colm[py, px] = raycolor
        farm = np.zeros((h, w), np.float32); farm[py, px] = far_dist
        near = near + ((hit > 0) & (farm > far_thresh) & (np.sqrt(((gtm - colm) ** 2).sum(-1)) < 0.1))
    opm = np.zeros((h, w), np.float32); opm[py, px] = opacity
    final = (hit > 0) * near * (opm > opacity_thresh) > 0                                # :544-546
    ray_of = -np.ones((h, w), np.int64); ray_of[py, px] = np.arange(len(px))
    return ray_of[final]                                                                 # boolean indexing = row-major order


def probe_hole(frames, order, pix, bg, h, w, far_thresh, opacity_thresh, prob_mul):
    """frames: list of dicts of per-ray arrays (+ 'gt'); order: frame ids as visited.  Returns (xyz, embedding, color, dir, conf) with the
    reference's accumulation, including its quirk: `add_conf = cat([add_conf, new]) * prob_mul` rescales the EARLIER frames' entries again."""
    xyz, emb, col, dr, conf = (np.zeros((0, 3), np.float32), np.zeros((0, 32), np.float32), np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32),
                               np.zeros((0, 1), np.float32))
    for i in order:
        f = frames[i]
        sel = frame_selection(pix, f["ray_mask"], f["coarse_raycolor"], f["ray_max_far_dist"][:, 0], f["ray_max_shading_opacity"][:, 0], f["gt"], bg, h, w,
                              far_thresh, opacity_thresh)
        xyz = np.concatenate([xyz, f["ray_max_sample_loc_w"][sel]])
        conf = (np.concatenate([conf, f["shading_avg_conf"][sel]]) * np.float32(prob_mul)).astype(np.float32)
        col = np.concatenate([col, f["shading_avg_color"][sel]])
        dr = np.concatenate([dr, f["shading_avg_dir"][sel]])
        emb = np.concatenate([emb, f["shading_avg_embedding"][sel]])
    return xyz, emb, col, dr, conf
