"""Drop-in module surface (NeuralPoints 14-tuple, NeuralPointsRayMarching output dict) vs goldens / oracle."""
import os
import tempfile

import numpy as np
import pytest
import torch

from tests.golden_io import load_render, torch_inputs

pytestmark = pytest.mark.gpu


def _build(tag):
    from hybridneuralrendering_amd import scenes
    from hybridneuralrendering_amd.modules import NeuralPoints, NeuralPointsRayMarching, find_blend_function, find_render_function, find_tone_map
    from hybridneuralrendering_amd.aggregator import PointAggregator
    d = load_render(tag)
    dev = torch.device("cuda:0")
    opt = scenes.default_opt(**d["opt"])
    ti = torch_inputs(d, dev)
    ckpt = {"neural_points.xyz": ti["xyz"].cpu(), "neural_points.points_embeding": ti["emb"].cpu(), "neural_points.points_conf": ti["conf"].cpu(),
            "neural_points.points_dir": ti["pdir"].cpu(), "neural_points.points_color": ti["color"].cpu()}
    with tempfile.NamedTemporaryFile(suffix=".pth", delete=False) as f:
        torch.save(ckpt, f.name)
        path = f.name
    npts = NeuralPoints(opt.point_features_dim, int(ti["xyz"].shape[0]), opt, dev, checkpoint=path).to(dev)
    os.unlink(path)
    agg = PointAggregator(opt)
    agg.load_state_dict(d["sd"], strict=True)
    agg = agg.to(dev)
    net = NeuralPointsRayMarching(tonemap_func=find_tone_map("off"), render_func=find_render_function("radiance"),
                                  blend_func=find_blend_function("alpha"), aggregator=agg, neural_points=npts, opt=opt,
                                  num_pos_freqs=opt.num_pos_freqs, num_viewdir_freqs=opt.num_viewdir_freqs)
    return d, ti, opt, npts, net, dev


def _inputs(d, ti, dev):
    near, far = d["near_far"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return dict(campos=ti["campos"], raydir=ti["raydir"], bg_color=ti["bg_color"], camrotc2w=ti["camrotc2w"],
                pixel_idx=t(d["pix"].astype(np.float32))[None], near=torch.tensor([[[near]]], device=dev, dtype=torch.float32),
                far=torch.tensor([[[far]]], device=dev, dtype=torch.float32), h=torch.tensor([48], device=dev), w=torch.tensor([64], device=dev),
                intrinsic=t(d["intrinsic"])[None], c2w=t(d["c2w"])[None], c2w_nearest=ti["c2w_nearest"], images_nearest=ti["images_nearest"],
                campos_nearest=ti["campos_nearest"], intrinsic_nearest=ti["intrinsic_nearest"], vid_angle_nearest=torch.zeros(1, 4, device=dev),
                frame_weight_nearest=torch.ones(1, 4, device=dev))


def test_neural_points_14_tuple_matches_oracle_gather():
    from oracle import render_oracle as ro
    d, ti, opt, npts, net, dev = _build("scannet_small")
    inp = _inputs(d, ti, dev)
    out = npts({k: inp[k] for k in ("pixel_idx", "camrotc2w", "campos", "near", "far", "h", "w", "intrinsic", "raydir")})
    assert len(out) == 14
    (s_color, s_Rw2c, s_dir, s_conf, s_emb, s_pers, s_xyz, s_mask, s_loc, s_loc_w, s_dirs, ray_mask, vsize, gvs) = out
    tc = torch_inputs(d)
    g = ro.gather_points(tc["xyz"], tc["emb"], tc["conf"], tc["pdir"], tc["color"], torch.from_numpy(d["q_sample_pidx"])[None],
                         tc["camrotc2w"], tc["campos"])
    np.testing.assert_array_equal(s_mask.cpu().numpy(), g["sample_pnt_mask"].numpy())
    np.testing.assert_array_equal(s_color.cpu().numpy(), g["sampled_color"].numpy())
    np.testing.assert_array_equal(s_dir.cpu().numpy(), g["sampled_dir"].numpy())
    np.testing.assert_array_equal(s_conf.cpu().numpy(), g["sampled_conf"].numpy())
    np.testing.assert_array_equal(s_emb.cpu().numpy(), g["sampled_embedding"].numpy())
    np.testing.assert_array_equal(s_xyz.cpu().numpy(), g["sampled_xyz"].numpy())
    np.testing.assert_allclose(s_pers.cpu().numpy(), g["sampled_xyz_pers"].numpy(), rtol=2e-6, atol=1e-6)
    np.testing.assert_array_equal(s_loc_w[0].cpu().numpy(), d["q_sample_loc_w"])
    np.testing.assert_array_equal(ray_mask[0].cpu().numpy(), d["q_ray_mask"])
    assert s_dirs.shape == s_loc_w.shape and s_loc.shape == s_loc_w.shape
    assert torch.equal(s_Rw2c.cpu(), torch.eye(3)) and vsize is opt.vsize and gvs == 0


@pytest.mark.parametrize("tag", ["scannet_small", "synth_small"])
def test_ray_marching_module_output_dict_matches_reference(tag):
    d, ti, opt, npts, net, dev = _build(tag)
    out = net(**_inputs(d, ti, dev))
    for k in ("coarse_raycolor", "coarse_raycolor_patch", "coarse_point_opacity", "coarse_is_background", "ray_mask", "weight",
              "blend_weight", "conf_coefficient", "queried_shading", "blur_predictor"):
        assert k in out
    tol = dict(rtol=0, atol=2e-4)
    for k in ("coarse_raycolor", "coarse_point_opacity", "coarse_is_background", "weight", "blend_weight", "conf_coefficient", "queried_shading"):
        assert tuple(out[k].shape) == d[k].shape, (k, out[k].shape, d[k].shape)
        np.testing.assert_allclose(out[k].cpu().numpy(), d[k], **tol)
    np.testing.assert_array_equal(out["ray_mask"].cpu().numpy(), d["ray_mask"])


def test_install_rebinds_reference_globals():
    import sys
    import types
    from hybridneuralrendering_amd import modules
    # a stand-in package tree with the reference's module names (the reference itself is absent on the GPU box)
    for name in ("models", "models.neural_points", "models.neural_points.neural_points", "models.neural_points_volumetric_model"):
        sys.modules.setdefault(name, types.ModuleType(name))
    vol = modules.install()
    assert vol.NeuralPoints is modules.NeuralPoints and vol.PointAggregator is modules.PointAggregator
    assert vol.NeuralPointsRayMarching is modules.NeuralPointsRayMarching and vol.ray_march is modules.ray_march
    assert sys.modules["models.neural_points.neural_points"].lighting_fast_querier_w is modules.Q.lighting_fast_querier
    for name in ("models", "models.neural_points", "models.neural_points.neural_points", "models.neural_points_volumetric_model"):
        sys.modules.pop(name, None)


def test_train_mode_module_gives_reference_gradients_on_its_parameters():
    """Drop-in surface in train mode: loss.backward() through NeuralPointsRayMarching fills .grad of the reference-named
    nn.Parameters (neural_points.points_*, aggregator.*) with the values torch autograd produced on the reference."""
    from tests.golden_io import load_train
    import tests.test_modules_gpu as me
    d = load_train("scannet_small")
    orig = me.load_render
    me.load_render = lambda tag: d                      # same scene + weights, train-mode options and ray batch
    try:
        d, ti, opt, npts, net, dev = _build("scannet_small")
    finally:
        me.load_render = orig
    assert opt.is_train == 1
    net.train()
    inp = _inputs(d, ti, dev)
    out = net(**inp, tmid=torch.from_numpy(d["tmid"]).to(dev))
    rows = np.nonzero(d["q_ray_mask"])[0]
    assert out["coarse_raycolor"].shape == (1, len(rows), 3) and out["coarse_raycolor"].requires_grad
    np.testing.assert_allclose(out["coarse_raycolor"].detach().cpu().numpy(), d["coarse_raycolor"], rtol=0, atol=2e-4)
    gt = torch.from_numpy(d["gt"]).to(dev)[:, rows]
    eps = float(d["zero_epsilon"])
    val = torch.clamp(out["conf_coefficient"], eps, 1 - eps)
    loss = torch.nn.functional.mse_loss(out["coarse_raycolor"], gt) + 1e-4 * torch.mean(torch.log(val) + torch.log(1 - val))
    np.testing.assert_allclose(loss.item(), d["loss"][0], rtol=2e-5)
    loss.backward()
    got = {"neural_points." + k: getattr(npts, k).grad for k in ("points_embeding", "points_conf", "points_dir", "points_color")}
    for k, prm in net.aggregator.named_parameters():
        if prm.grad is not None:
            got["aggregator." + k] = prm.grad
    assert set(got) == set(d["grad"])
    for k, ref in d["grad"].items():
        r = ref.numpy().astype(np.float64)
        x = got[k].detach().cpu().numpy().astype(np.float64).reshape(r.shape)
        tol = 1.5e-3 if k.startswith("neural_points.") else 3e-4        # see tests/test_train_gpu.py for the fp32-noise yardstick
        assert np.abs(x - r).max() <= tol * np.abs(r).max(), k
    assert npts.xyz.grad is None


def test_standalone_ray_march_matches_reference_formula():
    """modules.ray_march (hnr_ray_march) vs the torch restatement of diff_ray_marching.py:508-557."""
    from hybridneuralrendering_amd.modules import ray_march, radiance_render, alpha_blend
    from oracle import render_oracle as ro
    g = torch.Generator(device="cpu").manual_seed(2)
    R, SR = 777, 24
    feats = torch.cat([torch.nn.functional.softplus(torch.randn((1, R, SR, 1), generator=g)) * 30, torch.rand((1, R, SR, 3), generator=g)], -1)
    rd = torch.rand((1, R, SR), generator=g) * 0.02
    rv = torch.rand((1, R, SR), generator=g) > 0.3
    bg = torch.tensor([[0.2, 0.5, 1.0]])
    ref = ro.ray_march(rd, rv, feats, bg)
    out = ray_march(rd.cuda(), rv.cuda(), feats.cuda(), radiance_render, alpha_blend, bg.cuda())
    assert len(out) == 7
    for got, want in ((out[0], ref["ray_color"]), (out[2], ref["opacity"]), (out[3], ref["acc_transmission"]), (out[4], ref["blend_weight"]),
                      (out[5], ref["background_transmission"])):
        assert got.shape == want.shape
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=2e-6)
    assert torch.equal(out[1].cpu(), feats[..., 1:])


def test_prune_and_grow_rebuild_the_grid_and_match_the_oracle():
    """NeuralPoints.prune / grow_points (neural_points.py:350-402) change the cloud; the cached voxel grid must follow:
    the query after each edit is bit-identical to the oracle on the edited cloud."""
    from oracle import query_oracle as qo
    d, ti, opt, npts, net, dev = _build("scannet_small")
    inp = _inputs(d, ti, dev)
    o = d["opt"]
    near, far = d["near_far"]

    def oracle_query(xyz):
        hp = qo.hyperparameters(xyz, o["vsize"], o["vscale"], o["kernel_size"], o["ranges"], o["radius_limit_scale"])
        g = qo.OracleGrid(xyz, hp["origin"], hp["cell"], hp["dims"], o["query_size"], o["P"], o["max_o"])
        return g.query(d["c2w"][:3, 3], d["raydir"], qo.tmid_table(float(near), float(far), o["z_depth_dim"]), o["SR"], o["K"], hp["radius2"],
                       o["kernel_size"])

    def hip_query():
        out = npts({k: inp[k] for k in ("pixel_idx", "camrotc2w", "campos", "near", "far", "h", "w", "intrinsic", "raydir")})
        return out[7], out[11]                                   # sample_pnt_mask is [1,R',SR,K]; ray_mask [1,R]

    n0 = npts.xyz.shape[0]
    with torch.no_grad():
        npts.points_conf[0, ::3, 0] = 0.01                       # every third point falls below the threshold
    npts.prune(0.1)
    assert npts.xyz.shape[0] == n0 - (n0 + 2) // 3
    ref = oracle_query(npts.xyz.detach().cpu().numpy())
    m, rm = hip_query()
    np.testing.assert_array_equal(rm[0].cpu().numpy(), ref["ray_mask"])
    np.testing.assert_array_equal(m[0].cpu().numpy(), ref["sample_pidx"] >= 0)
    # grow: add a slab of new points in front of the camera
    g = torch.Generator().manual_seed(4)
    add = 3000
    cam = torch.from_numpy(d["c2w"][:3, 3]) + torch.from_numpy(d["c2w"][:3, 2]) * 0.25
    add_xyz = (cam[None] + (torch.rand((add, 3), generator=g) - 0.5) * torch.tensor([0.2, 0.2, 0.01])).to(dev)
    npts.grow_points(add_xyz, torch.zeros(add, 32, device=dev), torch.rand((add, 3), generator=g).to(dev), torch.zeros(add, 3, device=dev),
                     torch.ones(add, 1, device=dev))
    assert npts.xyz.shape[0] == n0 - (n0 + 2) // 3 + add and npts.points_embeding.shape[1] == npts.xyz.shape[0]
    ref2 = oracle_query(npts.xyz.detach().cpu().numpy())
    m2, rm2 = hip_query()
    np.testing.assert_array_equal(rm2[0].cpu().numpy(), ref2["ray_mask"])
    np.testing.assert_array_equal(m2[0].cpu().numpy(), ref2["sample_pidx"] >= 0)
    assert int((ref2["sample_pidx"] >= n0 - (n0 + 2) // 3).sum()) > 100          # the new points are actually found
    out = net(**inp)                                               # and the fused render runs on the edited cloud
    assert torch.isfinite(out["coarse_raycolor"]).all()


def test_reference_format_checkpoint_round_trip():
    """{iter}_net_ray_marching.pth = state_dict of NeuralPointsRayMarching with the reference's key names
    (models/base_model.py:91-108, neural_points.py:244-289): keys, save, load into a fresh module, same render."""
    d, ti, opt, npts, net, dev = _build("scannet_small")
    sd = net.state_dict()
    want = {"neural_points." + k for k in ("xyz", "points_embeding", "points_conf", "points_dir", "points_color")} | \
           {"aggregator." + k for k in d["sd"].keys()}
    assert set(sd.keys()) == want, set(sd.keys()) ^ want
    assert sd["neural_points.points_embeding"].shape == (1, npts.xyz.shape[0], 32) and sd["neural_points.points_conf"].shape[-1] == 1
    with tempfile.NamedTemporaryFile(suffix="_net_ray_marching.pth", delete=False) as f:
        torch.save({k: v.cpu() for k, v in sd.items()}, f.name)
        path = f.name
    from hybridneuralrendering_amd.modules import NeuralPoints, NeuralPointsRayMarching, find_blend_function, find_render_function, find_tone_map
    from hybridneuralrendering_amd.aggregator import PointAggregator
    npts2 = NeuralPoints(opt.point_features_dim, int(npts.xyz.shape[0]), opt, dev, checkpoint=path).to(dev)
    net2 = NeuralPointsRayMarching(tonemap_func=find_tone_map("off"), render_func=find_render_function("radiance"),
                                   blend_func=find_blend_function("alpha"), aggregator=PointAggregator(opt).to(dev), neural_points=npts2, opt=opt,
                                   num_pos_freqs=opt.num_pos_freqs, num_viewdir_freqs=opt.num_viewdir_freqs)
    missing, unexpected = net2.load_state_dict(torch.load(path, map_location=dev), strict=False)      # the shell loads with strict=False
    os.unlink(path)
    assert not missing and not unexpected
    inp = _inputs(d, ti, dev)
    a, b = net(**inp), net2(**inp)
    assert torch.equal(a["coarse_raycolor"], b["coarse_raycolor"])


def test_prob_outputs_match_reference():
    """opt.prob == 1: the hole-probing outputs of NeuralPointsRayMarching.forward (:392-416) vs the imported reference
    (tests/golden/render_scannet_small_prob.npz; same scene / weights / rays as render_scannet_small.npz)."""
    import tests.test_modules_gpu as me
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "render_scannet_small_prob.npz"))
    d0 = load_render("scannet_small")
    d0["opt"] = dict(d0["opt"], prob=1)
    orig = me.load_render
    me.load_render = lambda tag: d0
    try:
        d, ti, opt, npts, net, dev = _build("scannet_small")
    finally:
        me.load_render = orig
    out = net(**_inputs(d, ti, dev))
    np.testing.assert_allclose(out["coarse_raycolor"].cpu().numpy(), z["coarse_raycolor"], rtol=0, atol=2e-4)
    for k, tol in (("ray_max_shading_opacity", 2e-4), ("ray_max_sample_loc_w", 0.0), ("ray_max_far_dist", 1e-6), ("shading_avg_color", 2e-6),
                   ("shading_avg_dir", 2e-6), ("shading_avg_conf", 2e-6), ("shading_avg_embedding", 2e-6)):
        got, want = out[k].cpu().numpy(), z[k]
        assert got.shape == want.shape, (k, got.shape, want.shape)
        # the arg-max sample can differ where two opacities agree to rounding: allow a handful of such rays
        bad = np.abs(got - want).reshape(got.shape[1], -1).max(-1) > max(tol, 1e-7)
        assert bad.sum() <= 3, (k, int(bad.sum()), float(np.abs(got - want).max()))
