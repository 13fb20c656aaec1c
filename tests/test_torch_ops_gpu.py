"""torch.ops.hnr.* (csrc/torch_ops/hnr_torch.cpp, TORCH_LIBRARY(hnr)) against the ctypes route over the same C ABI: the registered ops must return
the same bits -- they are the same library calls behind a dispatcher schema (SURVEY 8b "C-ABI / op layer")."""
import numpy as np
import pytest
import torch

from tests.golden_io import load_render, load_train, torch_inputs

pytestmark = pytest.mark.gpu


def _render_setup(tag):
    from hybridneuralrendering_amd import scenes
    from hybridneuralrendering_amd.aggregator import PointAggregator
    from hybridneuralrendering_amd.render import HybridRenderer, PointCloud
    d = load_render(tag)
    dev = torch.device("cuda:0")
    opt = scenes.default_opt(**d["opt"])
    agg = PointAggregator(opt)
    agg.load_state_dict(d["sd"], strict=True)
    agg = agg.to(dev)
    ti = torch_inputs(d, dev)
    return d, ti, opt, agg, PointCloud(ti["xyz"], ti["emb"], ti["conf"], ti["pdir"], ti["color"]), HybridRenderer(opt, agg, dev)


def test_ops_are_registered_with_schemas():
    from hybridneuralrendering_amd import torch_ops
    ops = torch_ops.load()
    for name in ("grid_build", "grid_free", "march_query", "render_forward", "render_train"):
        assert getattr(ops, name).default._schema.name == "hnr::" + name
    with pytest.raises(Exception):                       # a CPU tensor has no kernel registered: the dispatcher refuses, nothing falls back
        ops.grid_build(torch.zeros(4, 3), [0., 0., 0.], [1., 1., 1.], [2, 2, 2], [3, 3, 3], 4, 10)


@pytest.mark.parametrize("tag", ["scannet_small", "synth_small"])
def test_render_forward_op_equals_the_ctypes_path_bit_for_bit(tag):
    from hybridneuralrendering_amd import torch_ops
    d, ti, opt, agg, cloud, rnd = _render_setup(tag)
    near, far = d["near_far"]
    w2c = torch.inverse(ti["c2w_nearest"][0])
    args = (cloud, ti["raydir"][0], ti["campos"][0], ti["camrotc2w"][0], ti["bg_color"][0], near, far, ti["c2w_nearest"][0], ti["campos_nearest"][0],
            ti["intrinsic_nearest"][0], ti["images_nearest"][0])
    ref = rnd.render_rays(*args, w2c_nearest=w2c)
    got = torch_ops.render_forward(rnd, *args, w2c_nearest=w2c)
    torch.cuda.synchronize()
    assert int(got["status"][0]) == 0
    for k in ("coarse_raycolor", "coarse_point_opacity", "coarse_is_background", "ray_mask", "decoded", "ray_nsamp", "counts"):
        assert torch.equal(got[k], ref[k]), k
    # and the golden of the imported reference, like the ctypes path (tests/test_render_gpu.py)
    np.testing.assert_allclose(got["coarse_raycolor"].cpu().numpy(), d["full_coarse_raycolor"][0], rtol=0, atol=2e-4)
    np.testing.assert_array_equal(got["ray_mask"].cpu().numpy(), d["q_ray_mask"])


def test_grid_and_query_ops_equal_the_ctypes_path():
    from hybridneuralrendering_amd import torch_ops, querier as Q
    d, ti, opt, agg, cloud, rnd = _render_setup("scannet_small")
    ops = torch_ops.load()
    grid, hp = rnd.querier._grid_for(cloud.xyz[None])
    near, far = d["near_far"]
    tmid = rnd.querier._tmid_for(float(near), float(far), opt.z_depth_dim, ti["raydir"].shape[1], cloud.xyz.device)
    r2 = float(np.float32(hp[0] ** 2))
    ref = Q.march_query(grid, ti["campos"][0].reshape(3), ti["raydir"][0], tmid, opt.SR, opt.K, np.float32(hp[0] ** 2), opt.kernel_size, pad=True)
    radius_limit_np, _, ranges_np, _, _, scaled_vsize_np, scaled_vdim_np = hp[:7]
    h = ops.grid_build(cloud.xyz, [float(v) for v in ranges_np[:3]], [float(v) for v in scaled_vsize_np], [int(v) for v in scaled_vdim_np],
                       [int(v) for v in opt.query_size], int(opt.P), int(opt.max_o))
    try:
        for handle in (h, torch_ops._handle(grid)):
            pidx, loc, nsamp, mask, counts = ops.march_query(handle, ti["campos"][0].reshape(3), ti["raydir"][0], tmid, int(opt.SR), int(opt.K), r2,
                                                             [int(k) for k in opt.kernel_size], True, 0)
            assert torch.equal(pidx, ref["sample_pidx"]) and torch.equal(loc, ref["sample_loc_w"]) and torch.equal(mask, ref["ray_mask"])
            assert torch.equal(nsamp, ref["ray_nsamp"]) and torch.equal(counts[:7], ref["counts"][:7])
    finally:
        ops.grid_free(h)
    rows = np.nonzero(d["q_ray_mask"])[0]
    np.testing.assert_array_equal(pidx.cpu().numpy()[rows], d["q_sample_pidx"])


def test_render_train_op_has_the_backward_registered_and_equals_the_ctypes_autograd_function():
    """hnr::render_train with its C++ autograd formula: same forward outputs, same gradients as train.render_train (the ctypes autograd.Function over the
    same two library calls) -- bit for bit where the library sums in a fixed order, to the order of a few float-atomic sums elsewhere -- and within the
    usual tolerance of the imported reference's gradients."""
    from hybridneuralrendering_amd import scenes, torch_ops
    from hybridneuralrendering_amd.aggregator import PointAggregator
    from hybridneuralrendering_amd.render import HybridRenderer
    from hybridneuralrendering_amd.train import TrainPath, render_train, drop_lut
    d = load_train("scannet_small")
    dev = torch.device("cuda:0")
    opt = scenes.default_opt(**d["opt"])
    agg = PointAggregator(opt)
    agg.load_state_dict(d["sd"], strict=True)
    agg = agg.to(dev)
    ti = torch_inputs(d, dev)
    rnd = HybridRenderer(opt, agg, dev)
    near, far = d["near_far"]
    tmid = torch.from_numpy(d["tmid"]).to(dev)
    gt = torch.from_numpy(d["gt"][0]).to(dev)
    mk = lambda t: t.clone().requires_grad_(True)

    def loss_of(out):
        m = out["ray_mask"] > 0
        cc = out["conf_coefficient"][m]
        val = torch.clamp(cc, float(d["zero_epsilon"]), 1 - float(d["zero_epsilon"]))
        return torch.nn.functional.mse_loss(out["coarse_raycolor"][m], gt[m]) + 1e-4 * torch.mean(torch.log(val) + torch.log(1 - val))
    args = (ti["raydir"][0], ti["campos"][0], ti["camrotc2w"][0], ti["bg_color"][0], near, far, ti["c2w_nearest"][0], ti["campos_nearest"][0],
            ti["intrinsic_nearest"][0], ti["images_nearest"][0])
    # ctypes autograd.Function
    l1 = [mk(ti[k]) for k in ("emb", "conf", "pdir", "color")]
    agg.zero_grad(set_to_none=True)
    o1 = render_train(TrainPath(rnd), agg, ti["xyz"], *l1, *args, tmid=tmid)
    loss_of(o1).backward()
    g1 = [t.grad.clone() for t in l1]
    w1 = {n: q.grad.clone() for n, q in agg.named_parameters() if q.grad is not None}
    # registered op
    l2 = [mk(ti[k]) for k in ("emb", "conf", "pdir", "color")]
    agg.zero_grad(set_to_none=True)
    o2 = torch_ops.render_train(rnd, agg, ti["xyz"], *l2, *args, tmid=tmid, drop_lut=drop_lut(opt, ti["raydir"].shape[1], dev))
    assert o2["coarse_raycolor"].requires_grad and o2["conf_coefficient"].requires_grad and not o2["decoded"].requires_grad
    for k in ("coarse_raycolor", "conf_coefficient", "ray_mask", "sample_pidx", "decoded", "coarse_point_opacity"):
        assert torch.equal(o2[k].detach(), o1[k].detach()), k
    loss_of(o2).backward()
    for a, b, k in zip(l2, g1, ("emb", "conf", "pdir", "color")):
        assert a.grad.shape == b.shape and torch.equal(a.grad, b), k
    w2 = {n: q.grad for n, q in agg.named_parameters() if q.grad is not None}
    assert sorted(w2) == sorted(w1)
    for n in w1:
        sc = float(w1[n].abs().max())
        assert float((w2[n] - w1[n]).abs().max()) <= 2e-6 * max(sc, 1e-30), n
    # the reference's golden gradients (same tolerance class as tests/test_train_gpu.py)
    ref = d["grad"]["neural_points.points_embeding"].numpy()
    err = np.abs(l2[0].grad.cpu().numpy().reshape(ref.shape) - ref).max() / np.abs(ref).max()
    assert err < 1.5e-3, err


def test_render_train_op_traces_forward_and_backward_without_running_a_kernel():
    """FakeTensorMode through hnr::render_train and its backward: the autograd formula is made of dispatcher ops with shape functions
    (render_train_fwd / render_train_bwd), so the whole step is traceable; torch.compile(backend="eager", fullgraph=True) captures a function
    that calls the op without a graph break and returns the bits of the eager call."""
    from torch._subclasses import FakeTensorMode
    from hybridneuralrendering_amd import torch_ops
    ops = torch_ops.load()
    with FakeTensorMode():
        c = lambda *s, dt=torch.float32: torch.empty(s, device="cuda", dtype=dt)
        R, SR, N = 64, 24, 300
        emb = c(1, N, 32).requires_grad_(True)
        ins = [c(N, 3), emb, c(1, N, 1), c(1, N, 3), c(1, N, 3), c(3), c(3, 3), c(R, 3), c(R, 400), c(3), None, None, None, None, None]
        ws = [c(4, 4).requires_grad_(True) for _ in range(44)]
        out = ops.render_train(0, ins, ws, None, None, SR, [3, 3, 3], 0.001, 0.008, 1, 0, 0.01, 0)
        (out[0].sum() + out[12].sum()).backward()
        assert tuple(emb.grad.shape) == (1, N, 32) and ws[0].grad.shape == (4, 4) and ws[20].grad is None      # (no views: no image-branch gradient)
    d, ti, opt, agg, cloud, rnd = _render_setup("scannet_small")
    near, far = d["near_far"]
    grid, hp = rnd.querier._grid_for(cloud.xyz[None])
    tmid = rnd.querier._tmid_for(float(near), float(far), opt.z_depth_dim, ti["raydir"].shape[1], cloud.xyz.device)
    h = torch_ops._handle(grid)
    args = (ti["campos"][0].reshape(3), ti["raydir"][0], tmid, int(opt.SR), int(opt.K), float(np.float32(hp[0] ** 2)), [int(k) for k in opt.kernel_size], True, 0)

    def f(campos, raydir, tm):
        pidx, loc, nsamp, mask, counts = torch.ops.hnr.march_query(h, campos, raydir, tm, args[3], args[4], args[5], args[6], args[7], args[8])
        return loc * 2.0, mask
    want = f(args[0], args[1], args[2])
    got = torch.compile(f, backend="eager", fullgraph=True)(args[0], args[1], args[2])
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
