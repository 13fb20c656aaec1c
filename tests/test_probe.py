"""Hole probing / point growing selection (/root/reference/run/train_ft.py:450-569): the CPU restatement against the
golden produced by the reference function, and the device implementation (hnr_probe_select + growth.probe_hole) against both."""
import os

import numpy as np
import pytest
import torch

from tests.golden_io import GOLD

KEYS = ("ray_mask", "coarse_raycolor", "ray_max_sample_loc_w", "ray_max_far_dist", "ray_max_shading_opacity", "shading_avg_color", "shading_avg_dir",
        "shading_avg_conf", "shading_avg_embedding")


def _load():
    z = np.load(os.path.join(GOLD, "probe_hole.npz"))
    frames = []
    for f in range(2):
        d = {k: z["f%d_%s" % (f, k)] for k in KEYS}
        d["gt"] = z["f%d_gt" % f]
        frames.append(d)
    return z, frames


@pytest.mark.parametrize("tag,far", [("far0", 0.0), ("far", 0.05)])
def test_probe_oracle_matches_reference_function(tag, far):
    from oracle import probe_oracle as po
    z, frames = _load()
    h, w = (int(v) for v in z["hw"])
    got = po.probe_hole(frames, [int(i) for i in z[tag + "_order"]], z["pix"], z["bg"], h, w, far, 0.7, 0.4)
    assert got[0].shape[0] > 50
    for name, g in zip(("xyz", "embedding", "color", "dir", "conf"), got):
        np.testing.assert_array_equal(g, z["%s_%s" % (tag, name)])


@pytest.mark.gpu
@pytest.mark.parametrize("tag,far", [("far0", 0.0), ("far", 0.05)])
def test_device_probe_hole_matches_reference_function(tag, far):
    from hybridneuralrendering_amd import growth
    z, frames = _load()
    h, w = (int(v) for v in z["hw"])
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    pix = t(z["pix"])[None]
    outs = []
    for i in z[tag + "_order"]:
        f = frames[int(i)]
        out = {k: t(f[k])[None] for k in KEYS}                       # [1, R, C] like the reference's output dict (ray_mask [1, R])
        outs.append((out, pix, t(f["gt"]), t(z["bg"])))
    add = growth.probe_hole(outs, h, w, far_thresh=far, opacity_thresh=0.7, prob_mul=0.4)
    for name, g in zip(("xyz", "embedding", "color", "dir", "conf"), add):
        np.testing.assert_array_equal(g.cpu().numpy(), z["%s_%s" % (tag, name)])


@pytest.mark.gpu
def test_probe_hole_then_grow_points_then_query():
    """The maintenance loop without the reference's process restart (run/train_ft.py:926-952): probe -> grow_points -> the next query
    sees the new points (the grid is rebuilt for the new cloud version)."""
    from hybridneuralrendering_amd import growth, scenes
    from hybridneuralrendering_amd.modules import NeuralPoints
    dev = torch.device("cuda:0")
    sc = scenes.make_scene("scene0241", 30000, 9, w=64, h=48)
    opt = sc.opt
    opt.load_points = 0
    npts = NeuralPoints(32, 30000, opt, dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    npts.set_points(t(sc.xyz), t(sc.emb), points_color=t(sc.color), points_dir=t(sc.dir), points_conf=t(sc.conf))      # scene attributes are [1, N, C]
    n0 = npts.xyz.shape[0]
    z, frames = _load()
    h, w = (int(v) for v in z["hw"])
    f = frames[0]
    out = {k: t(f[k])[None] for k in KEYS}
    add = growth.probe_hole([(out, t(z["pix"])[None], t(f["gt"]), t(z["bg"]))], h, w, far_thresh=0.0, opacity_thresh=0.7, prob_mul=0.4)
    assert add[0].shape[0] > 10
    # place the new points inside the scene so that the query can find them
    add_xyz = t(sc.xyz[:add[0].shape[0]]) + 0.001
    npts.grow_points(add_xyz, add[1], add[2], add[3], add[4])
    assert npts.xyz.shape[0] == n0 + add[0].shape[0] and npts.points_conf.shape[1] == npts.xyz.shape[0]
    pixg = scenes.pixel_grid(sc.w, sc.h)
    rays = t(scenes.camera_rays(pixg, sc.intrinsic, sc.c2w))
    res = npts.querier.query_points(t(pixg)[None], None, npts.xyz[None], None, sc.h, sc.w, sc.intrinsic, sc.near, sc.far, rays[None], t(sc.c2w[:3, 3])[None],
                                    t(sc.c2w[:3, :3])[None])
    assert int((res[0] >= n0).sum()) > 0                                 # some neighbour lists now name grown points
