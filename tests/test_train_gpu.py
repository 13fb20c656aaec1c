"""GPU parity of the training step (forward in train mode + hand-written HIP backward) against
 (a) the gradients torch autograd produced on the imported reference (tests/golden/train_scannet_small.npz) and
 (b) the CPU oracle's autograd on a fresh seeded batch,
plus unit tests of the two backward GEMMs against torch fp64.
Tolerances: gradients are sums over up to ~1e5 fp32 terms in a different order than the CPU run; each tensor is compared
with atol = GRAD_ATOL * max|ref| and rtol = GRAD_RTOL."""
import numpy as np
import pytest
import torch

from tests.golden_io import load_train, torch_inputs

pytestmark = pytest.mark.gpu

# Measured yardstick (tests/test_render_oracle.py::test_fp32_gradient_noise...): the reference's own fp32 CPU gradients differ
# from an fp64 run of the same graph by up to 2.4e-4 x max|g| (points_embeding; 6e-5 in l2) -- a point gradient is a short sum
# of terms that each went through ~12 fp32 layers and the ill-conditioned alpha-compositing derivative -- and by <= 4e-5 for
# the weight gradients (long sums, the noise averages out).  The HIP path is held to a small multiple of that noise.
GRAD_RTOL = 2e-3
TOL_POINTS = dict(max=1.5e-3, l2=4e-4)       # x max|ref|, relative l2
TOL_WEIGHTS = dict(max=3e-4, l2=1e-4)


def _loss(out, gt, zero_epsilon):
    """The shipped loss terms (models/base_rendering_model.py:1113-1118, :1228-1240) on the HIP outputs."""
    m = out["ray_mask"] > 0
    lc = torch.nn.functional.mse_loss(out["coarse_raycolor"][m], gt[m])
    cc = out["conf_coefficient"][m]                       # the reference's tensor holds the valid rays only
    val = torch.clamp(cc, zero_epsilon, 1 - zero_epsilon)
    lz = torch.mean(torch.log(val) + torch.log(1 - val))
    return lc + 1e-4 * lz, lc, lz


def _setup(tag="scannet_small"):
    from hybridneuralrendering_amd import scenes
    from hybridneuralrendering_amd.aggregator import PointAggregator
    from hybridneuralrendering_amd.render import HybridRenderer
    from hybridneuralrendering_amd.train import TrainPath
    d = load_train(tag)
    dev = torch.device("cuda:0")
    opt = scenes.default_opt(**{k: v for k, v in d["opt"].items()})
    assert opt.is_train == 1
    agg = PointAggregator(opt)
    agg.load_state_dict(d["sd"], strict=True)
    agg = agg.to(dev)
    ti = torch_inputs(d, dev)
    rnd = HybridRenderer(opt, agg, dev)
    return d, ti, opt, agg, TrainPath(rnd)


def _leaves(ti):
    mk = lambda t: t.clone().requires_grad_(True)
    return mk(ti["emb"]), mk(ti["conf"]), mk(ti["pdir"]), mk(ti["color"])


def _check_grads(got, ref, what, tol_weights=None):
    bad = []
    for k, g in ref.items():
        r = g.detach().cpu().numpy().astype(np.float64)
        x = got[k].detach().cpu().numpy().astype(np.float64).reshape(r.shape)
        scale = np.abs(r).max()
        assert scale > 0, k
        tol = TOL_POINTS if k.startswith("neural_points.") else (tol_weights or TOL_WEIGHTS)
        if r.size == 1:
            tol = dict(max=1e-3, l2=1e-3)     # a lone scalar (aux_merge_weight_block.6.bias) is a sum of signed terms that nearly cancel
        err = np.abs(x - r) - GRAD_RTOL * np.abs(r)
        l2 = np.linalg.norm(x - r) / np.linalg.norm(r)
        print("%-12s %-45s max|ref| %.3e  max err / max|ref| %.2e  rel l2 %.2e" % (what, k, scale, np.abs(x - r).max() / scale, l2))
        if err.max() > tol["max"] * scale or l2 > tol["l2"]:
            bad.append((k, float(np.abs(x - r).max() / scale), float(l2)))
    assert not bad, bad


@pytest.mark.parametrize("tag", ["scannet_small", "synth_small", "scannet_small_nearest0"])
def test_train_step_matches_reference_gradients(tag):
    from hybridneuralrendering_amd.train import render_train
    d, ti, opt, agg, path = _setup(tag)
    emb, conf, pdir, color = _leaves(ti)
    near, far = d["near_far"]
    tmid = torch.from_numpy(d["tmid"]).to(emb.device)
    out = render_train(path, agg, ti["xyz"], emb, conf, pdir, color, ti["raydir"][0], ti["campos"][0], ti["camrotc2w"][0],
                       ti["bg_color"][0], near, far, ti["c2w_nearest"][0], ti["campos_nearest"][0], ti["intrinsic_nearest"][0],
                       ti["images_nearest"][0], tmid=tmid)
    rows = np.nonzero(d["q_ray_mask"])[0]
    np.testing.assert_array_equal(out["ray_mask"].cpu().numpy(), d["q_ray_mask"])
    np.testing.assert_array_equal(out["sample_pidx"].cpu().numpy()[rows], d["q_sample_pidx"])          # jittered per-ray depths
    np.testing.assert_allclose(out["coarse_raycolor"].detach().cpu().numpy(), d["full_coarse_raycolor"][0], rtol=0, atol=2e-4)
    np.testing.assert_allclose(out["conf_coefficient"].detach().cpu().numpy()[rows], d["conf_coefficient"][0], rtol=0, atol=1e-7)
    loss, lc, lz = _loss(out, torch.from_numpy(d["gt"][0]).to(emb.device), float(d["zero_epsilon"]))
    np.testing.assert_allclose([loss.item(), lc.item(), lz.item()], d["loss"], rtol=2e-5)
    loss.backward()
    got = {"neural_points.points_embeding": emb.grad, "neural_points.points_conf": conf.grad,
           "neural_points.points_dir": pdir.grad, "neural_points.points_color": color.grad}
    for k, prm in agg.named_parameters():
        if prm.grad is not None:
            got["aggregator." + k] = prm.grad
    # the parameters that receive a gradient are the reference's (use_nearest = 0: none for the image branch)
    assert sorted(k for k in got if k.startswith("aggregator.")) == d["grad_names"], set(got) ^ set(d["grad_names"])
    _check_grads(got, d["grad"], "vs reference")


@pytest.mark.parametrize("tag", ["scannet_small", "scannet_small_nearest0"])
def test_graph_free_train_step_matches_reference_gradients(tag):
    """train.train_step (forward, hnr_shipped_loss_rows, backward queued back to back: no autograd graph, no masked copies, no host read)
    against the same golden gradients as the autograd path, and the loss terms against the reference's."""
    from hybridneuralrendering_amd.train import train_step
    d, ti, opt, agg, path = _setup(tag)
    emb, conf, pdir, color = _leaves(ti)
    near, far = d["near_far"]
    tmid = torch.from_numpy(d["tmid"]).to(emb.device)
    gt = torch.from_numpy(d["gt"][0]).to(emb.device)
    agg.zero_grad(set_to_none=True)
    out, pg, ag = train_step(path, agg, ti["xyz"], emb, conf, pdir, color, ti["raydir"][0], ti["campos"][0], ti["camrotc2w"][0], ti["bg_color"][0],
                             near, far, ti["c2w_nearest"][0], ti["campos_nearest"][0], ti["intrinsic_nearest"][0], ti["images_nearest"][0], gt,
                             zero_epsilon=float(d["zero_epsilon"]), w_color=1.0, w_zero_one=1e-4, tmid=tmid)
    parts = out["loss"].cpu().numpy()
    np.testing.assert_allclose([parts[1], parts[2]], d["loss"][1:], rtol=2e-5)
    np.testing.assert_allclose(parts[0] - 1e-6, d["loss"][0], rtol=2e-5, atol=1e-7)              # (`+ 1e-6` of compute_losses, base_rendering_model.py:1198)
    assert int(parts[3]) == int(d["q_ray_mask"].sum())
    got = {"neural_points.points_embeding": emb.grad, "neural_points.points_conf": conf.grad,
           "neural_points.points_dir": pdir.grad, "neural_points.points_color": color.grad}
    for k, prm in agg.named_parameters():
        if prm.grad is not None:
            got["aggregator." + k] = prm.grad
    assert sorted(k for k in got if k.startswith("aggregator.")) == d["grad_names"], set(got) ^ set(d["grad_names"])
    _check_grads(got, d["grad"], "graph-free vs reference")
    # a second call accumulates like autograd does
    before = emb.grad.clone()
    train_step(path, agg, ti["xyz"], emb, conf, pdir, color, ti["raydir"][0], ti["campos"][0], ti["camrotc2w"][0], ti["bg_color"][0],
               near, far, ti["c2w_nearest"][0], ti["campos_nearest"][0], ti["intrinsic_nearest"][0], ti["images_nearest"][0], gt,
               zero_epsilon=float(d["zero_epsilon"]), tmid=tmid)
    assert torch.equal(emb.grad, before + before)


def test_train_step_matches_oracle_on_a_fresh_batch():
    """Different camera window, fresh jitter, random upstream gradient: HIP backward vs the CPU oracle's autograd."""
    from hybridneuralrendering_amd import scenes
    from hybridneuralrendering_amd.train import render_train
    from oracle import render_oracle as ro
    from oracle import query_oracle as qo
    d, ti, opt, agg, path = _setup()
    dev = ti["emb"].device
    rng = np.random.default_rng(5)
    patch = 28
    x0, y0 = 20, 9
    px, py = np.meshgrid(np.arange(x0, x0 + patch), np.arange(y0, y0 + patch), indexing="ij")
    pix = np.stack([px, py], axis=-1).reshape(-1, 2).astype(np.int32)
    raydir = scenes.camera_rays(pix, d["intrinsic"], d["c2w"])
    near, far = d["near_far"]
    o = d["opt"]
    tmid = qo.tmid_table(float(near), float(far), o["z_depth_dim"])[None].repeat(raydir.shape[0], 0)
    tmid = (tmid + rng.uniform(-0.3, 0.3, size=tmid.shape) * (far - near) / o["z_depth_dim"] * 0.5).astype(np.float32)
    gt = rng.uniform(0, 1, size=(1, raydir.shape[0], 3)).astype(np.float32)
    # oracle
    hp = qo.hyperparameters(d["xyz"], o["vsize"], o["vscale"], o["kernel_size"], o["ranges"], o["radius_limit_scale"])
    grid = qo.OracleGrid(d["xyz"], hp["origin"], hp["cell"], hp["dims"], o["query_size"], o["P"], o["max_o"])
    q = grid.query(d["c2w"][:3, 3], raydir, tmid, o["SR"], o["K"], hp["radius2"], o["kernel_size"])
    tc = torch_inputs(d)
    drop = ro.drop_patch_rays(int(o["dilation_setup"].split("_")[1]), int(o["dilation_setup"].split("_")[0]), o["drop_ratio"])
    _, losses, ref = ro.train_step(tc["xyz"], tc["emb"], tc["conf"], tc["pdir"], tc["color"], d["sd"], q, tc["campos"], tc["camrotc2w"],
                                   torch.from_numpy(raydir)[None], tc["bg_color"], tc["c2w_nearest"], tc["campos_nearest"],
                                   tc["intrinsic_nearest"], tc["images_nearest"], o["vsize"], torch.from_numpy(gt),
                                   float(d["zero_epsilon"]), drop)
    # HIP
    emb, conf, pdir, color = _leaves(ti)
    out = render_train(path, agg, ti["xyz"], emb, conf, pdir, color, torch.from_numpy(raydir).to(dev), ti["campos"][0],
                       ti["camrotc2w"][0], ti["bg_color"][0], near, far, ti["c2w_nearest"][0], ti["campos_nearest"][0],
                       ti["intrinsic_nearest"][0], ti["images_nearest"][0], tmid=torch.from_numpy(tmid).to(dev))
    np.testing.assert_array_equal(out["ray_mask"].cpu().numpy(), q["ray_mask"])
    loss, lc, lz = _loss(out, torch.from_numpy(gt[0]).to(dev), float(d["zero_epsilon"]))
    np.testing.assert_allclose([loss.item(), lc.item(), lz.item()], losses, rtol=2e-5)
    loss.backward()
    got = {"neural_points.points_embeding": emb.grad, "neural_points.points_conf": conf.grad,
           "neural_points.points_dir": pdir.grad, "neural_points.points_color": color.grad}
    for k, prm in agg.named_parameters():
        if prm.grad is not None:
            got["aggregator." + k] = prm.grad
    # A hidden unit whose pre-activation is within rounding of 0 can sit on different sides of the LeakyReLU kink in the two
    # fp32 forwards (slope 1 vs 0.01 for that one element); on this batch that happens in color_feature_branch.0 and moves
    # its weight gradient by 7e-4 x max.  Weight gradients therefore get the point-gradient tolerance here; the reference
    # fixture above (no such element) holds them to TOL_WEIGHTS.
    _check_grads(got, ref, "vs oracle", tol_weights=TOL_POINTS)
    _, _, ref64 = ro.train_step(tc["xyz"], tc["emb"], tc["conf"], tc["pdir"], tc["color"], d["sd"], q, tc["campos"], tc["camrotc2w"],
                                torch.from_numpy(raydir)[None], tc["bg_color"], tc["c2w_nearest"], tc["campos_nearest"],
                                tc["intrinsic_nearest"], tc["images_nearest"], o["vsize"], torch.from_numpy(gt),
                                float(d["zero_epsilon"]), drop, dtype=torch.float64)
    _check_grads(got, ref64, "vs fp64", tol_weights=TOL_POINTS)


def test_train_step_on_a_batch_that_hits_nothing():
    """The camera stands far outside the cloud: zero valid samples run through every kernel of the step (device-side row counts of 0 -- the
    weight-stationary input gradient and the weight-gradient kernels among them).  The reference's loss for such a batch is the regulariser alone
    (colour loss set to 0, base_rendering_model.py:1144); all gradients are exact zeros and nothing hangs or reads out of bounds."""
    from hybridneuralrendering_amd.train import train_step
    d, ti, opt, agg, path = _setup()
    dev = ti["emb"].device
    near, far = d["near_far"]
    raydir = ti["raydir"][0]
    campos = (ti["campos"][0] + 1000.0).contiguous()                       # a kilometre away: no ray meets the grid between near and far
    gt = torch.rand((raydir.shape[0], 3), device=dev)
    emb, conf, pdir, color = _leaves(ti)
    for _ in range(2):                                                     # twice: the second step reuses the buffers of the first
        out, pg, ag = train_step(path, agg, ti["xyz"], emb, conf, pdir, color, raydir.contiguous(), campos, ti["camrotc2w"][0], ti["bg_color"][0], near, far,
                                 ti["c2w_nearest"][0], ti["campos_nearest"][0], ti["intrinsic_nearest"][0], ti["images_nearest"][0], gt,
                                 zero_epsilon=float(d["zero_epsilon"]), assign_grads=False)
        torch.cuda.synchronize()
        path.check_status(out)
        assert int(out["ray_mask"].sum()) == 0 and int(out["counts"][6]) == 0
        assert torch.isfinite(out["loss"]).all()
        for k, g in list(pg.items()) + list(ag.items()):
            assert torch.isfinite(g).all() and float(g.abs().max()) == 0.0, k


def test_train_step_keeps_the_two_frame_weights_apart():
    """The item's scalar `frame_weight` multiplies loss_total (models/base_rendering_model.py:1204-1205); `frame_weight_nearest` [1,V] scales the
    per-view merge weights (models/aggregators/point_aggregators.py:1202-1203).  train_step takes them as two arguments (one
    argument used to feed both): loss and gradients against the oracle's autograd with both set, and the wrong shapes are refused."""
    from hybridneuralrendering_amd import scenes
    from hybridneuralrendering_amd.train import train_step
    from hybridneuralrendering_amd._lib import HnrError
    from oracle import render_oracle as ro
    from oracle import query_oracle as qo
    d, ti, opt, agg, path = _setup()
    dev = ti["emb"].device
    rng = np.random.default_rng(11)
    px, py = np.meshgrid(np.arange(31, 31 + 24), np.arange(14, 14 + 24), indexing="ij")
    pix = np.stack([px, py], axis=-1).reshape(-1, 2).astype(np.int32)
    raydir = scenes.camera_rays(pix, d["intrinsic"], d["c2w"])
    near, far = d["near_far"]
    o = d["opt"]
    tmid = qo.tmid_table(float(near), float(far), o["z_depth_dim"])[None].repeat(raydir.shape[0], 0).astype(np.float32)
    gt = rng.uniform(0, 1, size=(1, raydir.shape[0], 3)).astype(np.float32)
    fw, fwn = 0.7, np.array([[1.0, 0.35, 0.8, 0.55]], np.float32)
    hp = qo.hyperparameters(d["xyz"], o["vsize"], o["vscale"], o["kernel_size"], o["ranges"], o["radius_limit_scale"])
    grid = qo.OracleGrid(d["xyz"], hp["origin"], hp["cell"], hp["dims"], o["query_size"], o["P"], o["max_o"])
    q = grid.query(d["c2w"][:3, 3], raydir, tmid, o["SR"], o["K"], hp["radius2"], o["kernel_size"])
    tc = torch_inputs(d)
    drop = ro.drop_patch_rays(int(o["dilation_setup"].split("_")[1]), int(o["dilation_setup"].split("_")[0]), o["drop_ratio"])
    _, losses, ref = ro.train_step(tc["xyz"], tc["emb"], tc["conf"], tc["pdir"], tc["color"], d["sd"], q, tc["campos"], tc["camrotc2w"],
                                   torch.from_numpy(raydir)[None], tc["bg_color"], tc["c2w_nearest"], tc["campos_nearest"],
                                   tc["intrinsic_nearest"], tc["images_nearest"], o["vsize"], torch.from_numpy(gt),
                                   float(d["zero_epsilon"]), drop, frame_weight=fw, frame_weight_n=torch.from_numpy(fwn))
    emb, conf, pdir, color = _leaves(ti)
    agg.zero_grad(set_to_none=True)
    args = (path, agg, ti["xyz"], emb, conf, pdir, color, torch.from_numpy(raydir).to(dev), ti["campos"][0], ti["camrotc2w"][0], ti["bg_color"][0],
            near, far, ti["c2w_nearest"][0], ti["campos_nearest"][0], ti["intrinsic_nearest"][0], ti["images_nearest"][0], torch.from_numpy(gt[0]).to(dev))
    kw = dict(zero_epsilon=float(d["zero_epsilon"]), w_color=1.0, w_zero_one=1e-4, tmid=torch.from_numpy(tmid).to(dev))
    out, pg, ag = train_step(*args, frame_weight=fw, frame_weight_nearest=torch.from_numpy(fwn).to(dev), **kw)
    parts = out["loss"].cpu().numpy()
    np.testing.assert_allclose(parts[0], losses[0] + 1e-6 * fw, rtol=3e-5, atol=1e-7)          # (`+ 1e-6` of compute_losses, base_rendering_model.py:1198, before the scaling)
    np.testing.assert_allclose([parts[1], parts[2]], losses[1:], rtol=3e-5)
    got = {"neural_points.points_embeding": emb.grad, "neural_points.points_conf": conf.grad,
           "neural_points.points_dir": pdir.grad, "neural_points.points_color": color.grad}
    for k, prm in agg.named_parameters():
        if prm.grad is not None:
            got["aggregator." + k] = prm.grad
    _check_grads(got, ref, "two frame weights vs oracle", tol_weights=TOL_POINTS)
    # the per-view weights really reached the merge: without them the image branch's gradients differ
    emb2, conf2, pdir2, color2 = _leaves(ti)
    agg.zero_grad(set_to_none=True)
    args2 = (path, agg, ti["xyz"], emb2, conf2, pdir2, color2) + args[7:]
    train_step(*args2, frame_weight=fw, **kw)
    k0 = "aux_merge_weight_block.0.weight"
    assert not torch.allclose(dict(agg.named_parameters())[k0].grad, got["aggregator." + k0], rtol=1e-3, atol=0.0)
    # refused: the scalar where the per-view vector belongs, a device tensor / a vector as the scalar
    with pytest.raises(HnrError):
        train_step(*args, frame_weight_nearest=torch.tensor([fw], device=dev), **kw)
    with pytest.raises(HnrError):
        train_step(*args, frame_weight=torch.tensor([fw], device=dev), **kw)
    with pytest.raises(HnrError):
        train_step(*args, frame_weight=torch.from_numpy(fwn), **kw)


@pytest.mark.parametrize("M,n_cols,n_keys,two", [(5000, 256, 300, False), (20000, 48, 1500, True), (17, 48, 5, True), (100000, 256, 40000, False)])
def test_sort_and_segment_sum_equal_index_add(M, n_cols, n_keys, two):
    """hnr_sort_rows_by_key + hnr_segment_sum_rows == torch index_add over the rows with key >= 0."""
    import ctypes
    from hybridneuralrendering_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(M + n_cols)
    keys = torch.randint(-1, n_keys, (M,), generator=g, dtype=torch.int32).cuda()
    A = torch.randn((M, n_cols), generator=g).cuda()
    B = torch.randn((M, n_cols), generator=g).cuda() if two else None
    ks, perm = torch.empty_like(keys), torch.empty_like(keys)
    sb = int(L.hnr_sort_rows_scratch_bytes(M))
    scratch = torch.empty((sb,), dtype=torch.uint8, device="cuda")
    _lib.check(L.hnr_sort_rows_by_key(_lib.ptr(keys), M, _lib.ptr(ks), _lib.ptr(perm), _lib.ptr(scratch), sb, _lib.stream()), "sort")
    order = torch.sort(keys.cpu().to(torch.int64), stable=True)
    assert torch.equal(ks.cpu().to(torch.int64), order.values) and torch.equal(perm.cpu().to(torch.int64), order.indices)
    dst = torch.zeros((n_keys, n_cols + 4), device="cuda")
    _lib.check(L.hnr_segment_sum_rows(_lib.ptr(A), n_cols, _lib.ptr(B) if two else None, n_cols if two else 0, _lib.ptr(ks), _lib.ptr(perm), M,
                                      n_cols, _lib.ptr(dst), n_cols + 4, _lib.stream()), "segment sum")
    src = (A + B) if two else A
    m = keys >= 0
    ref = torch.zeros((n_keys, n_cols), dtype=torch.float64, device="cuda").index_add_(0, keys[m].long(), src[m].double())
    assert (dst[:, :n_cols].double() - ref).abs().max().item() < 1e-4
    assert torch.all(dst[:, n_cols:] == 0)


def test_point_buffer_gradients_are_bit_identical_run_to_run():
    """The gradients of points_embeding / points_conf / points_dir / points_color (the transposed gather of
    models/neural_points/neural_points.py:709-720) are summed per touched point in a FIXED order -- rows sorted by point with a
    stable radix sort, one wave per point (hnr_segment_sum_rows_det) -- not with float atomics: repeated steps give the same bits."""
    from hybridneuralrendering_amd.train import render_train
    d, ti, opt, agg, path = _setup("scannet_small")
    near, far = d["near_far"]
    tmid = torch.from_numpy(d["tmid"]).to(ti["emb"].device)
    gt = torch.from_numpy(d["gt"][0]).to(ti["emb"].device)
    runs = []
    for _ in range(3):
        emb, conf, pdir, color = _leaves(ti)
        agg.zero_grad(set_to_none=True)
        out = render_train(path, agg, ti["xyz"], emb, conf, pdir, color, ti["raydir"][0], ti["campos"][0], ti["camrotc2w"][0],
                           ti["bg_color"][0], near, far, ti["c2w_nearest"][0], ti["campos_nearest"][0], ti["intrinsic_nearest"][0],
                           ti["images_nearest"][0], tmid=tmid)
        loss, _, _ = _loss(out, gt, float(d["zero_epsilon"]))
        loss.backward()
        runs.append([g.grad.clone() for g in (emb, conf, pdir, color)])
    for other in runs[1:]:
        for a, b, name in zip(runs[0], other, ("points_embeding", "points_conf", "points_dir", "points_color")):
            assert torch.equal(a, b), name


@pytest.mark.parametrize("tag", ["scannet_small", "scannet_small_nearest0"])
def test_train_step_is_bit_identical_run_to_run_with_the_side_streams(tag):
    """The training calls fork the reference-view CNN, their buffer clears and weight packs onto streams of the library (csrc/render_train.hip) and
    join them before returning: every gradient that is not an atomic sum by design must come out with the same bits in every repetition of a
    step (a missing dependency between the streams shows up as last-bit differences in a few rows -- tools/race_probe.py)."""
    from hybridneuralrendering_amd.train import train_step
    d, ti, opt, agg, path = _setup(tag)
    near, far = d["near_far"]
    tmid = torch.from_numpy(d["tmid"]).to(ti["emb"].device)
    gt = torch.from_numpy(d["gt"][0]).to(ti["emb"].device)
    atomic = ("aux_block", "alpha_branch", "color_final", "aux_merge_weight_block.6")      # weight gradients summed with float atomics
    ref = None
    for it in range(12):
        emb, conf, pdir, color = _leaves(ti)
        out, pg, ag = train_step(path, agg, ti["xyz"], emb, conf, pdir, color, ti["raydir"][0], ti["campos"][0], ti["camrotc2w"][0], ti["bg_color"][0], near, far,
                                 ti["c2w_nearest"][0], ti["campos_nearest"][0], ti["intrinsic_nearest"][0], ti["images_nearest"][0], gt,
                                 zero_epsilon=float(d["zero_epsilon"]), tmid=tmid, assign_grads=False)
        cur = {("points." + k): v.clone() for k, v in pg.items()}
        cur.update({k: v.clone() for k, v in ag.items() if not k.startswith(atomic)})
        cur["coarse_raycolor"] = out["coarse_raycolor"].clone()
        if ref is None:
            ref = cur
            continue
        bad = [k for k in ref if not torch.equal(cur[k], ref[k])]
        assert not bad, (it, bad)


@pytest.mark.parametrize("n_keys,max_rows,n_cols,two", [(300, 40, 256, True), (64, 700, 256, False), (5, 5000, 64, True), (1, 1, 8, False), (2000, 3, 128, True)])
def test_point_major_segment_sum_equals_index_add(n_keys, max_rows, n_cols, two):
    """hnr_segment_sum_rows_csr (the training step's per-point sums: a point-major row list in ANY order, no sort) == torch index_add, for segments on
    every path of the kernel -- up to a few rows, hundreds (rank sort in LDS), and more than 2 048 rows (the repeated-minimum fallback) -- and its
    optional second matrix and maximum; a second call on a differently shuffled list gives the same bits (the order of the list must not matter)."""
    from hybridneuralrendering_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(n_keys * 7 + max_rows)
    cnt = rng.integers(0, max_rows + 1, size=n_keys).astype(np.int32)
    cnt[rng.integers(0, n_keys)] = max_rows                         # at least one segment of the largest size
    start = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int32)
    M = int(cnt.sum())
    rows = rng.permutation(M + 37)[:M].astype(np.int32)              # distinct row indices, arbitrary order inside every segment
    A = torch.randn((M + 37, n_cols), generator=torch.Generator().manual_seed(1)).cuda()
    A2 = torch.randn((M + 37, 8), generator=torch.Generator().manual_seed(2)).cuda() if two else None
    key_of = np.repeat(np.arange(n_keys), cnt)
    ref = torch.zeros((n_keys, n_cols), dtype=torch.float64).index_add_(0, torch.from_numpy(key_of).long(), A.cpu().double()[rows.astype(np.int64)])
    ref2 = torch.zeros((n_keys, 8), dtype=torch.float64).index_add_(0, torch.from_numpy(key_of).long(), A2.cpu().double()[rows.astype(np.int64)]) if two else None

    def run(row_list):
        dst = torch.full((n_keys, n_cols + 4), 9.0, device="cuda")
        dst2 = torch.full((n_keys, 8), 9.0, device="cuda") if two else None
        amax = torch.zeros((1,), dtype=torch.int32, device="cuda")
        rl, st, ct = (torch.from_numpy(x).cuda() for x in (row_list, start, cnt))
        _lib.check(L.hnr_segment_sum_rows_csr(_lib.ptr(A), n_cols, _lib.ptr(rl), _lib.ptr(st), _lib.ptr(ct), n_cols, n_keys, _lib.ptr(dst), n_cols + 4,
                                              _lib.ptr(A2) if two else None, 8 if two else 0, 8 if two else 0, _lib.ptr(dst2) if two else None, 8 if two else 0,
                                              _lib.ptr(amax), _lib.stream()), "hnr_segment_sum_rows_csr")
        return dst, dst2, amax

    dst, dst2, amax = run(rows)
    scale = max(1.0, float(ref.abs().max()))
    assert (dst[:, :n_cols].cpu().double() - ref).abs().max().item() < 2e-6 * scale * max(1, max_rows) ** 0.5
    assert torch.all(dst[:, n_cols:] == 9.0)
    if two:
        assert (dst2.cpu().double() - ref2).abs().max().item() < 2e-6 * max(1.0, float(ref2.abs().max())) * max(1, max_rows) ** 0.5
    # the published maximum has the exponent of the true one (csrc/hnr_common.h absmax_publish)
    true_max = float(dst[:, :n_cols].abs().max())
    got = float(amax.view(torch.float32).item())
    assert true_max == 0.0 or (got <= true_max and np.frexp(got)[1] == np.frexp(true_max)[1])
    # any order of a segment's rows gives the same bits
    shuffled = rows.copy()
    for k in range(n_keys):
        seg = shuffled[start[k]:start[k] + cnt[k]]
        rng.shuffle(seg)
    d_b, d2_b, _ = run(shuffled)
    assert torch.equal(d_b, dst) and (not two or torch.equal(d2_b, dst2))


def _blur_kernels(dev, n=6, ks=5, seed=5):
    g = torch.Generator().manual_seed(seed)
    k = torch.rand((n, ks, ks), generator=g) ** 3
    return (k / k.sum(dim=(1, 2), keepdim=True)).to(dev)[None]


def test_graph_free_step_with_the_blur_module_equals_the_autograd_form():
    """train_step(blur_kernels=...) = forward -> hnr_blur_select -> loss kernels -> hnr_blur_select_bwd -> backward, queued back to back, against
    render_train + blur.blur_update_output + losses.shipped_loss + loss.backward() (models/mvs_points_volumetric_model.py:145-146 between the
    render and compute_losses): same launches, so the gradients agree bit for bit.  Also: the touched-point list the forward call leaves on
    the device is torch.unique(sample_pidx >= 0)."""
    from hybridneuralrendering_amd.train import train_step, render_train, TrainPath
    from hybridneuralrendering_amd.blur import blur_update_output
    from hybridneuralrendering_amd.losses import shipped_loss
    d, ti, opt, agg, path = _setup()
    dev = ti["emb"].device
    near, far = d["near_far"]
    tmid = torch.from_numpy(d["tmid"]).to(dev)
    gt = torch.from_numpy(d["gt"][0]).to(dev)
    pn, ps = (int(x) for x in d["opt"]["dilation_setup"].split("_")[:2])
    assert pn * ps * pn * ps == gt.shape[0]
    kern = _blur_kernels(dev)
    args = (ti["raydir"][0], ti["campos"][0], ti["camrotc2w"][0], ti["bg_color"][0], near, far, ti["c2w_nearest"][0], ti["campos_nearest"][0],
            ti["intrinsic_nearest"][0], ti["images_nearest"][0])
    # autograd form
    emb, conf, pdir, color = _leaves(ti)
    agg.zero_grad(set_to_none=True)
    o = render_train(path, agg, ti["xyz"], emb, conf, pdir, color, *args, tmid=tmid)
    col = blur_update_output(o["coarse_raycolor"][None], gt[None], kern, pn, ps)[0]
    loss, parts = shipped_loss(col, o["conf_coefficient"], gt, o["ray_mask"], float(d["zero_epsilon"]), 1.0, 1e-4, frame_weight=0.7, conf_rows=True)
    loss.backward()
    ref = dict(emb=emb.grad.clone(), conf=conf.grad.clone(), pdir=pdir.grad.clone(), color=color.grad.clone())
    refw = {n: q.grad.clone() for n, q in agg.named_parameters() if q.grad is not None}
    # graph-free form
    emb2, conf2, pdir2, color2 = _leaves(ti)
    agg.zero_grad(set_to_none=True)
    out, pg, ag = train_step(path, agg, ti["xyz"], emb2, conf2, pdir2, color2, *args, gt, zero_epsilon=float(d["zero_epsilon"]), w_color=1.0, w_zero_one=1e-4,
                             frame_weight=0.7, tmid=tmid, blur_kernels=kern, patch_num=pn, patch_size=ps)
    assert torch.equal(out["loss"], parts)
    assert torch.equal(out["blurred_raycolor"].reshape(col.shape), col.detach())
    for k, t in (("emb", emb2), ("conf", conf2), ("pdir", pdir2), ("color", color2)):
        assert torch.equal(t.grad, ref[k]), k
    for n, q in agg.named_parameters():
        if n in refw:
            # (a few narrow weight gradients -- alpha branch, final colour, the 64 -> 1 merge-weight layer, the conv pyramid -- are float-atomic sums:
            # equal up to the order of their additions; every other gradient is a fixed-order sum)
            sc = float(refw[n].abs().max())
            assert float((q.grad - refw[n]).abs().max()) <= 2e-6 * max(sc, 1e-30), n
    # the blur module really changed the colours (some patch selected a non-identity kernel)
    assert not torch.equal(out["blurred_raycolor"], out["coarse_raycolor"])
    ids, cnt = TrainPath.touched_points(out["_saved"])
    n = int(cnt.item())
    want = torch.unique(out["sample_pidx"][out["sample_pidx"] >= 0])
    assert n == want.numel() and torch.equal(ids[:n].long(), want.long())


def test_chained_c5_step_matches_reference_gradients():
    """BASELINE config C5 as the reference chains it (models/mvs_points_volumetric_model.py:135-152): train-mode forward on a dilated patch batch ->
    blur_update_output (models/base_rendering_model.py:677-745) -> compute_losses with the item's frame weight (:1205-1206) -> backward.
    tests/golden/train_c5_small.npz = the imported reference's blurred colours, loss_total and all 48 + 4 gradients of that chain
    (make_golden.py::gen_train_c5); train.train_step(blur_kernels=..., frame_weight=...) -- the production step bench.py times -- against them."""
    from hybridneuralrendering_amd.train import train_step
    d, ti, opt, agg, path = _setup("c5_small")
    dev = ti["emb"].device
    emb, conf, pdir, color = _leaves(ti)
    near, far = d["near_far"]
    tmid = torch.from_numpy(d["tmid"]).to(dev)
    gt = torch.from_numpy(d["gt"][0]).to(dev)
    pn, ps = (int(v) for v in d["patch"])
    fw = float(d["frame_weight"])
    kern = torch.from_numpy(d["blur_kernels"]).to(dev)[None]
    agg.zero_grad(set_to_none=True)
    out, pg, ag = train_step(path, agg, ti["xyz"], emb, conf, pdir, color, ti["raydir"][0], ti["campos"][0], ti["camrotc2w"][0], ti["bg_color"][0],
                             near, far, ti["c2w_nearest"][0], ti["campos_nearest"][0], ti["intrinsic_nearest"][0], ti["images_nearest"][0], gt,
                             zero_epsilon=float(d["zero_epsilon"]), w_color=1.0, w_zero_one=1e-4, frame_weight=fw, tmid=tmid,
                             blur_kernels=kern, patch_num=pn, patch_size=ps)
    np.testing.assert_array_equal(out["ray_mask"].cpu().numpy(), d["q_ray_mask"])
    np.testing.assert_allclose(out["coarse_raycolor"].cpu().numpy(), d["full_coarse_raycolor"][0], rtol=0, atol=2e-4)
    # the blur module's choice per patch and its colours are the reference's
    np.testing.assert_allclose(out["blurred_raycolor"].reshape(-1, 3).cpu().numpy(), d["blurred_raycolor"][0], rtol=0, atol=2e-4)
    assert not np.array_equal(d["blurred_raycolor"], d["full_coarse_raycolor"])
    parts = out["loss"].cpu().numpy()
    np.testing.assert_allclose([parts[1], parts[2]], d["loss"][1:], rtol=2e-5)
    assert abs(float(parts[0]) - float(d["loss_compute_losses"])) < 5e-6                       # (compute_losses' constant 1e-6 per colour item, :1198)
    assert int(parts[3]) == int(d["q_ray_mask"].sum())
    got = {"neural_points.points_embeding": emb.grad, "neural_points.points_conf": conf.grad,
           "neural_points.points_dir": pdir.grad, "neural_points.points_color": color.grad}
    for k, prm in agg.named_parameters():
        if prm.grad is not None:
            got["aggregator." + k] = prm.grad
    assert sorted(k for k in got if k.startswith("aggregator.")) == d["grad_names"], set(got) ^ set(d["grad_names"])
    _check_grads(got, d["grad"], "chained C5 step vs reference")


def test_reused_output_buffers_keep_autograd_grad_semantics():
    """TrainPath.reuse_outputs: every step overwrites the same gradient buffers.  .grad must never alias them (round-5 advice): after
    optimizer.zero_grad(set_to_none=False) a second step leaves g2 (not 2 x g2), and without zeroing it leaves g1 + g2; accumulate_grads=False hands
    out the buffers themselves and a later accumulating step refuses to add into them."""
    from hybridneuralrendering_amd.train import train_step
    from hybridneuralrendering_amd import HnrError
    d, ti, opt, agg, path = _setup()
    path.reuse_outputs = True
    dev = ti["emb"].device
    near, far = d["near_far"]
    tmid = torch.from_numpy(d["tmid"]).to(dev)
    gt = torch.from_numpy(d["gt"][0]).to(dev)
    gt2 = torch.rand_like(gt)
    emb, conf, pdir, color = _leaves(ti)
    leaves = dict(points_embeding=emb, points_conf=conf, points_dir=pdir, points_color=color)

    def step(g, **kw):
        return train_step(path, agg, ti["xyz"], emb, conf, pdir, color, ti["raydir"][0], ti["campos"][0], ti["camrotc2w"][0], ti["bg_color"][0], near, far,
                          ti["c2w_nearest"][0], ti["campos_nearest"][0], ti["intrinsic_nearest"][0], ti["images_nearest"][0], g,
                          zero_epsilon=float(d["zero_epsilon"]), tmid=tmid, **kw)

    def grads():
        out = {k: t.grad.clone() for k, t in leaves.items()}
        out.update({n: q.grad.clone() for n, q in agg.named_parameters() if q.grad is not None})
        return out
    _o, pg, ag = step(gt)
    g1 = grads()
    assert emb.grad.data_ptr() != pg["points_embeding"].data_ptr()
    assert all(q.grad.data_ptr() != ag[n].data_ptr() for n, q in agg.named_parameters() if n in ag)
    # accumulation without zeroing: g1 + g2
    step(gt2)
    acc = grads()
    # zero in place (the torch < 2.0 default and a common explicit choice), second step again: exactly g2
    for t in leaves.values():
        t.grad.zero_()
    agg.zero_grad(set_to_none=False)
    step(gt2)
    g2 = grads()
    for k in g1:
        # point gradients are fixed-order sums: exact; a few narrow weight gradients are float-atomic sums, equal up to the order of their additions
        want = g1[k] + g2[k]
        if k in leaves:
            assert torch.equal(acc[k], want), k
        else:
            assert float((acc[k] - want).abs().max()) <= 4e-6 * max(float(want.abs().max()), 1e-30), k
    assert float((g2["points_embeding"] - g1["points_embeding"]).abs().max()) > 0          # another target: another gradient
    # accumulate_grads=False: the step's own buffers, and an accumulating step afterwards refuses to add into them
    for t in leaves.values():
        t.grad = None
    agg.zero_grad(set_to_none=True)
    _o, pg, ag = step(gt2, accumulate_grads=False)
    assert emb.grad.data_ptr() == pg["points_embeding"].data_ptr()
    assert torch.equal(emb.grad.reshape(-1), g2["points_embeding"].reshape(-1))
    with pytest.raises(HnrError):
        step(gt)


def test_captured_train_step_replays_bit_identically_and_follows_its_inputs():
    """train.CapturedTrainStep: the step (forward, blur module, loss kernels, backward: ~150 launches on three queues) captured once in a hipGraph;
    a replay with the same inputs equals the eager train_step bit for bit, a replay with another ray batch / ground truth / frame weight equals the
    eager step on THOSE inputs (the launch sizes are capacity-fixed and read device counters), and a third replay of the first inputs returns the
    first bits again."""
    from hybridneuralrendering_amd.train import train_step, CapturedTrainStep
    d, ti, opt, agg, path = _setup()
    dev = ti["emb"].device
    near, far = d["near_far"]
    tmid = torch.from_numpy(d["tmid"]).to(dev)
    gt = torch.from_numpy(d["gt"][0]).to(dev)
    pn, ps = (int(x) for x in d["opt"]["dilation_setup"].split("_")[:2])
    kern = _blur_kernels(dev)
    raydir = ti["raydir"][0]
    # a second batch: the same rays mirrored in order, another target, another jitter
    perm = torch.arange(raydir.shape[0] - 1, -1, -1, device=dev)
    raydir_b, gt_b, tmid_b = raydir[perm].contiguous(), torch.rand_like(gt), tmid[perm].contiguous()
    emb, conf, pdir, color = _leaves(ti)

    def eager(rd, g, tm, fw):
        for t in (emb, conf, pdir, color):
            t.grad = None
        agg.zero_grad(set_to_none=True)
        out, pg, ag = train_step(path, agg, ti["xyz"], emb, conf, pdir, color, rd, ti["campos"][0], ti["camrotc2w"][0], ti["bg_color"][0], near, far,
                                 ti["c2w_nearest"][0], ti["campos_nearest"][0], ti["intrinsic_nearest"][0], ti["images_nearest"][0], g,
                                 zero_epsilon=float(d["zero_epsilon"]), frame_weight=fw, tmid=tm, blur_kernels=kern, patch_num=pn, patch_size=ps)
        return (out["loss"].clone(), out["coarse_raycolor"].clone(), {k: v.clone() for k, v in pg.items()}, {k: v.clone() for k, v in ag.items()})
    ref_a, ref_b = eager(raydir, gt, tmid, 0.7), eager(raydir_b, gt_b, tmid_b, 0.4)
    sample = dict(raydir=raydir, campos=ti["campos"][0], camrot=ti["camrotc2w"][0], bg_color=ti["bg_color"][0], c2w_nearest=ti["c2w_nearest"][0],
                  campos_nearest=ti["campos_nearest"][0], intrinsic_nearest=ti["intrinsic_nearest"][0], images_nearest=ti["images_nearest"][0],
                  gt_image=gt, tmid=tmid, blur_kernels=kern, frame_weight=0.7)
    cap = CapturedTrainStep(path, agg, ti["xyz"], emb, conf, pdir, color, sample, near, far, zero_epsilon=float(d["zero_epsilon"]),
                            patch_num=pn, patch_size=ps)

    def same(got, ref, what):
        out, pg, ag = got
        assert torch.equal(out["loss"], ref[0]), (what, out["loss"], ref[0])
        assert torch.equal(out["coarse_raycolor"], ref[1]), what
        for k in ref[2]:
            assert torch.equal(pg[k], ref[2][k]), (what, k)
        for k in ref[3]:
            sc = float(ref[3][k].abs().max())
            # float-atomic sums in a few narrow layers; a lone scalar (aux_merge_weight_block.6.bias) is a sum of signed terms that nearly cancel
            assert float((ag[k] - ref[3][k]).abs().max()) <= (1e-3 if ref[3][k].numel() == 1 else 2e-6) * max(sc, 1e-30), (what, k)
    same(cap.step(), ref_a, "replay A")
    same(cap.step(raydir=raydir_b, gt_image=gt_b, tmid=tmid_b, frame_weight=0.4), ref_b, "replay B")
    same(cap.step(raydir=raydir, gt_image=gt, tmid=tmid, frame_weight=0.7), ref_a, "replay A again")
    assert emb.grad is not None and torch.equal(emb.grad.reshape(-1, 32), cap.pg["points_embeding"])


def test_captured_train_step_draws_a_fresh_jitter_per_replay():
    """tmid=None at construction: the jittered depth tables are drawn inside the graph (torch.rand under capture); two replays use different
    tables (different samples -> different losses), like two eager steps do (query_point_indices_worldcoords.py:87)."""
    from hybridneuralrendering_amd.train import CapturedTrainStep
    d, ti, opt, agg, path = _setup()
    dev = ti["emb"].device
    near, far = d["near_far"]
    gt = torch.from_numpy(d["gt"][0]).to(dev)
    emb, conf, pdir, color = _leaves(ti)
    sample = dict(raydir=ti["raydir"][0], campos=ti["campos"][0], camrot=ti["camrotc2w"][0], bg_color=ti["bg_color"][0], c2w_nearest=ti["c2w_nearest"][0],
                  campos_nearest=ti["campos_nearest"][0], intrinsic_nearest=ti["intrinsic_nearest"][0], images_nearest=ti["images_nearest"][0], gt_image=gt)
    cap = CapturedTrainStep(path, agg, ti["xyz"], emb, conf, pdir, color, sample, near, far, zero_epsilon=float(d["zero_epsilon"]))
    l1 = cap.step()[0]["loss"].clone()
    l2 = cap.step()[0]["loss"].clone()
    assert torch.isfinite(l1).all() and torch.isfinite(l2).all() and not torch.equal(l1, l2)
    assert abs(float(l1[0]) - float(l2[0])) < 0.05 * abs(float(l1[0]))


def test_point_gradient_exchange_kernels_equal_the_torch_form():
    """parallel.PointGradExchange on the GPU: hnr_point_grad_pack / hnr_point_grad_apply (csrc/exchange.hip) against the torch-op form of the same
    class (what the gloo tests run on CPU tensors), bit for bit, on three ranks' records stacked by hand: overlapping touched sets, point 0 touched by
    one rank, reached only through the empty slots' conf gradient on another, and absent on the third; unequal valid-ray counts; an overflowing rank."""
    import os
    from hybridneuralrendering_amd import parallel
    dev = torch.device("cuda:0")
    N, cap = 20000, 700
    g = torch.Generator().manual_seed(3)

    def rank_data(r):
        n = (300, 650, 120)[r]
        ids = (torch.randperm(N - 1, generator=g)[:n] + 1).sort().values.to(torch.int32)
        if r == 0:
            ids[0] = 0
        bufs = [torch.zeros(N, 32), torch.zeros(N), torch.zeros(N, 3), torch.zeros(N, 3)]
        for b in bufs:
            b[ids.long()] = torch.randn((n,) + tuple(b.shape[1:]), generator=g)
        if r == 1:
            bufs[1][0] = 0.375                     # the empty slots' conf gradient on point 0, which this rank did not touch
        pad = torch.cat([ids, torch.full((50,), 7, dtype=torch.int32)])
        return [b.to(dev) for b in bufs], pad.to(dev), torch.tensor([n], dtype=torch.int64, device=dev), torch.tensor([float(900 + 411 * r)], device=dev)
    data = [rank_data(r) for r in range(3)]
    ex = parallel.PointGradExchange(cap)
    recs_hip = [ex.pack(b, i, c, nv) for b, i, c, nv in data]
    os.environ["HNR_EXCHANGE_TORCH"] = "1"
    try:
        recs_t = [ex.pack(b, i, c, nv) for b, i, c, nv in data]
        for a, b in zip(recs_hip, recs_t):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
        allr = torch.stack(recs_t).contiguous()
        want = []
        for r in range(3):
            bufs = [b.clone() for b in data[r][0]]
            tot, over = ex.apply(allr, bufs, r)
            want.append(bufs)
        assert float(tot) == 900 * 3 + 411 * 3 and float(over) == 0
    finally:
        del os.environ["HNR_EXCHANGE_TORCH"]
    for r in range(3):
        bufs = [b.clone() for b in data[r][0]]
        tot, over = ex.apply(allr, bufs, r)
        assert float(tot) == 900 * 3 + 411 * 3 and float(over) == 0
        for a, b, w0 in zip(bufs, want[r], want[0]):
            assert torch.equal(a, b) and torch.equal(a, w0)           # HIP == torch, and every rank holds the same bits
    # the weighted sum itself, against a dense fp64 evaluation
    n = [900.0, 1311.0, 1722.0]
    for k in range(4):
        dense = sum(data[r][0][k].double() * (n[r] / sum(n)) for r in range(3))
        assert float((want[0][k].double() - dense).abs().max()) < 1e-6
    assert abs(float(want[0][1][0]) - 0.375 * n[1] / sum(n) - float(data[0][0][1][0]) * n[0] / sum(n)) < 1e-6
    # an overflowing rank is flagged by both forms
    small = parallel.PointGradExchange(256)
    rec = small.pack(*data[1])
    assert float(rec[0, 2]) == 1.0 and float(rec[0, 0]) == 256.0


_CHAIN_AB = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from tests.test_train_gpu import _setup, _leaves
from hybridneuralrendering_amd.train import train_step
arrs = {}
for tag in ("scannet_small", "synth_small"):
    d, ti, opt, agg, path = _setup(tag)
    near, far = d["near_far"]
    tmid = torch.from_numpy(d["tmid"]).to(ti["emb"].device)
    gt = torch.from_numpy(d["gt"][0]).to(ti["emb"].device)
    emb, conf, pdir, color = _leaves(ti)
    out, pg, ag = train_step(path, agg, ti["xyz"], emb, conf, pdir, color, ti["raydir"][0], ti["campos"][0], ti["camrotc2w"][0], ti["bg_color"][0], near, far,
                             ti["c2w_nearest"][0], ti["campos_nearest"][0], ti["intrinsic_nearest"][0], ti["images_nearest"][0], gt,
                             zero_epsilon=float(d["zero_epsilon"]), tmid=tmid, assign_grads=False)
    arrs.update({tag + ".out." + k: out[k].detach().cpu().numpy() for k in out if torch.is_tensor(out[k])})
    arrs.update({tag + ".pg." + k: pg[k].detach().cpu().numpy() for k in pg})
    arrs.update({tag + ".ag." + k: ag[k].detach().cpu().numpy() for k in ag})
np.savez(sys.argv[2], **arrs)
'''


def test_weight_stationary_training_chain_agrees_with_the_layer_by_layer_kernel(tmp_path):
    """The training forward's per-neighbour chain runs in the render path's weight-stationary pipelined kernel in its activation-keeping form
    (chain_ws_kernel<8>, csrc/chain_ws.hip); HNR_TRAIN_CHAIN_WS=0 selects the layer-by-layer chain_kernel<4, 3>.  The two add a layer's k steps in
    different orders (chain_ws.hip, "Load schedule"), so they agree to fp32 rounding, not bit for bit: every output and gradient within 5e-6 of
    the tensor's maximum (measured <= 9e-7), and everything upstream of the chain (query, aggregation weights) identical.  The switch is read once per
    process, hence two child processes."""
    import os, subprocess, sys
    root = str(__import__("pathlib").Path(__file__).resolve().parents[1])
    script = tmp_path / "chain_ab.py"
    script.write_text(_CHAIN_AB)
    res = []
    for v in ("1", "0"):
        f = tmp_path / ("ws%s.npz" % v)
        p = subprocess.run([sys.executable, str(script), root, str(f)], capture_output=True, text=True, timeout=900, env=dict(os.environ, HNR_TRAIN_CHAIN_WS=v), cwd=root)
        assert p.returncode == 0, p.stderr[-2000:]
        res.append(np.load(f))
    a, b = res
    assert sorted(a.files) == sorted(b.files)
    upstream = ("ray_mask", "sample_pidx", "sample_loc_w", "ray_nsamp", "counts", "status", "weight", "conf_coefficient")
    for k in a.files:
        if ".out." in k and k.split(".out.")[-1] in upstream:
            assert np.array_equal(a[k], b[k]), k
            continue
        x, y = a[k].astype(np.float64), b[k].astype(np.float64)
        sc = max(np.abs(y).max(), 1e-30)
        assert np.abs(x - y).max() <= (1e-3 if x.size == 1 else 5e-6) * sc, (k, np.abs(x - y).max() / sc)
