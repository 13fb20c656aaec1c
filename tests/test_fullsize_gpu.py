"""BASELINE.json configs at their full sizes: C2 lego-800 (640 000 rays, 1.0 M points, SR 80, P 9, query +
composite only) and C4 scene0101 (4.0 M points, P 30, max_o 2 M, a 620x460 frame).  The CPU oracle cannot run these in
seconds, so the checks are (a) bit-exact agreement with the C oracle on a random subset of the rays (the grid is built over
ALL points on both sides) and (b) size-independent properties of the full outputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _query_setup(name, n_points, seed):
    from hybridneuralrendering_amd import scenes, querier as Q
    sc = scenes.make_scene(name, int(n_points), seed)
    opt = sc.opt
    dev = torch.device("cuda:0")
    xyz = torch.from_numpy(sc.xyz).to(dev)
    mn, mx = Q.points_bounds(xyz)
    rl, ranges_np, cell, dims, _ = Q.compute_hyperparameters(mn, mx, opt.vsize, opt.vscale, opt.kernel_size, opt.ranges, opt.radius_limit_scale)
    grid = Q.VoxelGrid(xyz, ranges_np[:3], cell, dims, opt.query_size, opt.P, opt.max_o)
    return sc, opt, dev, xyz, grid, (rl, ranges_np, cell, dims)


def _check_query_properties(res, xyz, r2, SR, K):
    pidx, loc, nsamp, mask, counts = res["sample_pidx"], res["sample_loc_w"], res["ray_nsamp"], res["ray_mask"], res["counts"]
    from hybridneuralrendering_amd._lib import CNT
    R = pidx.shape[0]
    valid = pidx >= 0
    # -1 padding is a suffix of every K-list, kept samples are a prefix of every ray
    assert not bool((valid[..., 1:] & ~valid[..., :-1]).any())
    slot = torch.arange(SR, device=pidx.device)[None, :]
    assert not bool((valid.any(-1) & (slot >= nsamp[:, None])).any())
    # counters agree with the tensors
    assert int(counts[CNT["SAMPLES"]]) == int(nsamp.sum())
    assert int(counts[CNT["NEIGHBOURS"]]) == int(valid.sum())
    assert int(counts[CNT["SAMPLES_VALID"]]) == int(valid.any(-1).sum())
    assert torch.equal(mask.bool(), valid.any(-1).any(-1))
    # every neighbour lies inside the radius, ids are unique inside a sample (checked on a slice to bound memory)
    rows = torch.arange(0, R, max(R // 20000, 1), device=pidx.device)
    p, l = pidx[rows], loc[rows]
    d = xyz[p.clamp(min=0).long()] - l[:, :, None, :]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    assert bool((d2[p >= 0] <= r2).all())
    srt = torch.sort(p, dim=-1).values
    assert not bool(((srt[..., 1:] == srt[..., :-1]) & (srt[..., 1:] >= 0)).any())


def _check_subset_against_oracle(sc, opt, hp, res, rays_np, tmid_np, n_sub, seed):
    from oracle import query_oracle as qo
    rl, ranges_np, cell, dims = hp
    g = qo.OracleGrid(sc.xyz, ranges_np[:3], cell, dims, opt.query_size, opt.P, opt.max_o)
    sel = np.sort(np.random.default_rng(seed).choice(rays_np.shape[0], size=n_sub, replace=False))
    o = g.query(sc.c2w[:3, 3], rays_np[sel], tmid_np, opt.SR, opt.K, np.float32(rl ** 2), opt.kernel_size, want_full=True)
    np.testing.assert_array_equal(res["sample_pidx"][sel].cpu().numpy(), o["full_pidx"])
    np.testing.assert_array_equal(res["sample_loc_w"][sel].cpu().numpy(), o["full_loc"])
    np.testing.assert_array_equal(res["ray_nsamp"][sel].cpu().numpy(), o["full_nsamp"])
    np.testing.assert_array_equal(res["ray_mask"][sel].cpu().numpy(), o["ray_mask"])


def test_c2_lego_800_query_and_composite_full_size():
    from hybridneuralrendering_amd import scenes, querier as Q, _lib
    sc, opt, dev, xyz, grid, hp = _query_setup("lego", 1.0e6, 1)
    assert (sc.w, sc.h, opt.SR, opt.P) == (800, 800, 80, 9)
    pix = scenes.pixel_grid(sc.w, sc.h, 0)
    rays_np = scenes.camera_rays(pix, sc.intrinsic, sc.c2w)
    assert rays_np.shape[0] == 640000
    rays = torch.from_numpy(rays_np).to(dev)
    campos = torch.from_numpy(sc.c2w[:3, 3].copy()).to(dev)
    tmid = Q.tmid_table(sc.near, sc.far, opt.z_depth_dim, device=dev)
    r2 = np.float32(hp[0] ** 2)
    res = Q.march_query(grid, campos, rays, tmid, opt.SR, opt.K, r2, opt.kernel_size)
    assert int(res["ray_mask"].sum()) > 50000                      # the object covers a good part of the frame
    _check_query_properties(res, xyz, float(r2), opt.SR, opt.K)
    _check_subset_against_oracle(sc, opt, hp, res, rays_np, tmid.cpu().numpy(), 3000, 5)
    res2 = Q.march_query(grid, campos, rays, tmid, opt.SR, opt.K, r2, opt.kernel_size)
    assert torch.equal(res["sample_pidx"], res2["sample_pidx"]) and torch.equal(res["sample_loc_w"], res2["sample_loc_w"])   # deterministic
    # composite on synthetic decoded features (sigma ~ softplus(N(0,1)), rgb ~ U[0,1])
    L = _lib.lib()
    R, SR, K = res["sample_pidx"].shape
    g = torch.Generator(device="cpu").manual_seed(3)
    dec = torch.cat([torch.nn.functional.softplus(torch.randn((R, SR, 1), generator=g)) * 40.0, torch.rand((R, SR, 3), generator=g)], -1).to(dev)
    dec = dec * (res["sample_pidx"][..., :1] >= 0)                  # the aggregate writes zeros where a sample has no neighbour
    camrot = torch.from_numpy(sc.c2w[:3, :3].copy()).to(dev)
    bg = torch.tensor([1.0, 1.0, 1.0], device=dev)

    def comp(d):
        col, opa, isbg, bw = (torch.empty((R, 3), device=dev), torch.empty((R, SR), device=dev), torch.empty((R,), device=dev),
                              torch.empty((R, SR), device=dev))
        _lib.check(L.hnr_composite(_lib.ptr(d.contiguous()), _lib.ptr(res["sample_loc_w"]), _lib.ptr(res["sample_pidx"]), _lib.ptr(res["ray_mask"]),
                                   None, _lib.ptr(campos), _lib.ptr(camrot), _lib.ptr(bg), R, SR, K, float(np.float32(opt.vsize[2])), 1,
                                   _lib.ptr(col), _lib.ptr(opa), _lib.ptr(isbg), _lib.ptr(bw), _lib.stream()), "hnr_composite")
        return col, opa, isbg, bw
    col, opa, isbg, bw = comp(dec)
    # telescoping: sum of blend weights + final transmittance = 1 (up to the 1e-10 the reference adds per factor)
    assert float((bw.sum(-1) + isbg - 1).abs().max()) < 1e-4
    assert float(col.min()) >= -1e-5 and float(col.max()) <= 1 + 1e-5
    miss = res["ray_mask"] == 0
    assert bool((col[miss] == 1).all()) and bool((isbg[miss] == 1).all()) and bool((opa[miss] == 0).all())
    # linear in the colours: composite(a * rgb, bg = 0) = a * composite(rgb, bg = 0)
    bg0 = torch.zeros(3, device=dev)
    bg, keep = bg0, bg
    c1 = comp(dec)[0]
    half = dec.clone(); half[..., 1:] *= 0.5
    c2 = comp(half)[0]
    assert float((c1 * 0.5 - c2).abs().max()) < 1e-6
    # against the torch restatement on a subset of valid rays
    from oracle import render_oracle as ro
    rows = torch.nonzero(res["ray_mask"])[:2000, 0]
    sl = ro.w2pers_samples(res["sample_loc_w"][rows].cpu()[None], torch.from_numpy(sc.c2w[:3, :3].copy())[None], torch.from_numpy(sc.c2w[:3, 3].copy())[None])
    rv = (res["sample_pidx"][rows][..., 0] >= 0).cpu()[None]
    rd = ro.ray_dist(sl, rv, float(np.float32(opt.vsize[2])), 1)
    m = ro.ray_march(rd, rv, dec[rows].cpu()[None], torch.zeros(1, 3))
    assert float((m["ray_color"][0] - c1[rows].cpu()).abs().max()) < 2e-5


def test_c4_scene0101_4m_points_frame_query_full_size():
    from hybridneuralrendering_amd import scenes, querier as Q
    sc, opt, dev, xyz, grid, hp = _query_setup("scene0101", 4.0e6, 3)
    assert (opt.P, opt.max_o, opt.SR) == (30, 2000000, 24)
    st = grid.stats
    assert st["n_points"] == 4000000 and st["n_dropped_voxels"] == 0 and 500000 < st["n_occ"] <= opt.max_o
    pix = scenes.pixel_grid(sc.w, sc.h, 10)
    rays_np = scenes.camera_rays(pix, sc.intrinsic, sc.c2w)
    assert rays_np.shape[0] == 285200
    rays = torch.from_numpy(rays_np).to(dev)
    campos = torch.from_numpy(sc.c2w[:3, 3].copy()).to(dev)
    tmid = Q.tmid_table(sc.near, sc.far, opt.z_depth_dim, device=dev)
    r2 = np.float32(hp[0] ** 2)
    res = Q.march_query(grid, campos, rays, tmid, opt.SR, opt.K, r2, opt.kernel_size)
    assert int(res["ray_mask"].sum()) > 0.99 * rays_np.shape[0]    # closed room: (almost) every ray finds neighbours
    _check_query_properties(res, xyz, float(r2), opt.SR, opt.K)
    _check_subset_against_oracle(sc, opt, hp, res, rays_np, tmid.cpu().numpy(), 2000, 7)
    # 8-way sharding of the frame (C4): the per-rank blocks give the same rows as the whole-frame launch
    from hybridneuralrendering_amd.parallel import shard_bounds
    for rank in (0, 3, 7):
        lo, hi = shard_bounds(rays.shape[0], 8, rank)
        part = Q.march_query(grid, campos, rays[lo:hi].contiguous(), tmid, opt.SR, opt.K, r2, opt.kernel_size)
        assert torch.equal(part["sample_pidx"], res["sample_pidx"][lo:hi]) and torch.equal(part["ray_mask"], res["ray_mask"][lo:hi])


def test_c3_scene0241_train_step_full_size():
    """Config C3 at its full size: 2.0 M points, one 56x56 = 3136-ray training batch (jittered depths, patch drop), forward +
    HIP backward vs the CPU oracle's autograd on the same batch (the oracle needs ~1 min for this one)."""
    from hybridneuralrendering_amd import scenes
    from hybridneuralrendering_amd.aggregator import PointAggregator
    from hybridneuralrendering_amd.render import HybridRenderer
    from hybridneuralrendering_amd.train import TrainPath, render_train
    from oracle import query_oracle as qo, render_oracle as ro
    sc = scenes.make_scene("scene0241", int(2.0e6), 2)
    opt = sc.opt
    opt.is_train = 1
    assert opt.dilation_setup == "7_8_1_8" and (opt.SR, opt.P, opt.max_o) == (24, 26, 610000)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    agg = PointAggregator(opt)
    with torch.no_grad():
        agg.alpha_branch[0].weight.mul_(30.0)
        agg.alpha_branch[0].bias.fill_(30.0)
    sd = {k: v.detach().clone() for k, v in agg.state_dict().items()}
    agg = agg.to(dev)
    rng = np.random.default_rng(3)
    px, py = np.meshgrid(np.arange(300, 356), np.arange(200, 256), indexing="ij")
    pix = np.stack([px, py], axis=-1).reshape(-1, 2).astype(np.int32)
    rays = scenes.camera_rays(pix, sc.intrinsic, sc.c2w)
    tm = qo.tmid_table(sc.near, sc.far, opt.z_depth_dim)
    tm = (tm[None].repeat(rays.shape[0], 0) + rng.uniform(-0.15, 0.15, size=(rays.shape[0], tm.shape[0])) * (sc.far - sc.near) / opt.z_depth_dim).astype(np.float32)
    gt = rng.uniform(0, 1, size=(1, rays.shape[0], 3)).astype(np.float32)
    hp = qo.hyperparameters(sc.xyz, opt.vsize, opt.vscale, opt.kernel_size, opt.ranges, opt.radius_limit_scale)
    og = qo.OracleGrid(sc.xyz, hp["origin"], hp["cell"], hp["dims"], opt.query_size, opt.P, opt.max_o)
    q = og.query(sc.c2w[:3, 3], rays, tm, opt.SR, opt.K, hp["radius2"], opt.kernel_size)
    c = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    t = lambda a: c(a).to(dev)
    _, losses, gref = ro.train_step(c(sc.xyz), c(sc.emb), c(sc.conf), c(sc.dir), c(sc.color), sd, q, c(sc.c2w[:3, 3])[None], c(sc.c2w[:3, :3])[None],
                                    c(rays)[None], c(sc.bg_color)[None], c(sc.c2w_nearest)[None], c(sc.c2w_nearest[:, :3, 3])[None],
                                    c(sc.intrinsic)[None], c(sc.images_nearest)[None], opt.vsize, c(gt), 1e-3, ro.drop_patch_rays(8, 7, opt.drop_ratio))
    leaves = [t(a).requires_grad_(True) for a in (sc.emb, sc.conf, sc.dir, sc.color)]
    for prm in agg.parameters():
        prm.requires_grad_(True)
    o = render_train(TrainPath(HybridRenderer(opt, agg, dev)), agg, t(sc.xyz), leaves[0], leaves[1], leaves[2], leaves[3], t(rays), t(sc.c2w[:3, 3]),
                     t(sc.c2w[:3, :3]), t(sc.bg_color), sc.near, sc.far, t(sc.c2w_nearest), t(sc.c2w_nearest[:, :3, 3]), t(sc.intrinsic),
                     t(sc.images_nearest), tmid=t(tm))
    np.testing.assert_array_equal(o["ray_mask"].cpu().numpy(), q["ray_mask"])
    m = o["ray_mask"] > 0
    val = torch.clamp(o["conf_coefficient"][m], 1e-3, 1 - 1e-3)
    loss = torch.nn.functional.mse_loss(o["coarse_raycolor"][m], t(gt[0])[m]) + 1e-4 * torch.mean(torch.log(val) + torch.log(1 - val))
    assert abs(loss.item() - losses[0]) < 2e-5 * abs(losses[0])
    loss.backward()
    got = {"neural_points.points_embeding": leaves[0].grad, "neural_points.points_conf": leaves[1].grad, "neural_points.points_dir": leaves[2].grad,
           "neural_points.points_color": leaves[3].grad}
    got.update({"aggregator." + k: v.grad for k, v in agg.named_parameters() if v.grad is not None})
    assert set(got) == set(gref)
    worst_w, worst_p = 0.0, 0.0
    for k, r in gref.items():
        r = r.numpy().astype(np.float64)
        x = got[k].detach().cpu().numpy().astype(np.float64).reshape(r.shape)
        if r.size == 1:
            continue
        e = float(np.abs(x - r).max() / np.abs(r).max())
        l2 = float(np.linalg.norm(x - r) / np.linalg.norm(r))
        if k.startswith("neural_points."):
            worst_p = max(worst_p, e)
            assert e < 4e-3 and l2 < 1e-3, (k, e, l2)     # deterministic per-point sums (no atomics): the same value every run
        else:
            worst_w = max(worst_w, e)
            # fp32 effects at 272 k rows: a LeakyReLU-kink flip (see test_train_gpu.py) moves a weight gradient by ~1e-3; the
            # merge-weight MLP's gradients are tiny (|g| ~ 1e-6) sums dominated by few samples, and a reprojected sample within
            # rounding of a pixel border reads the neighbouring pixel on one side (the forward tests allow the same): 3e-3 / 8e-4
            tol_e, tol_l2 = (8e-3, 3e-3) if "aux_merge_weight_block" in k else (2e-3, 1.5e-3)
            assert e < tol_e and l2 < tol_l2, (k, e, l2)
    print("C3 full size: %d rows; worst gradient error: weights %.1e, points %.1e of max" % (int(o["counts"][3]), worst_w, worst_p))
    # fp64 yardstick for the widest tolerances above (round-5 verdict: "would not notice a 0.3 % systematic error in a merge-weight gradient"): the same
    # graph in double precision on the CPU; the HIP gradients must be as close to it as the reference's own fp32 evaluation is (x 3), i.e. what the
    # tolerances absorb is fp32 rounding of the reference arithmetic, not an error of the kernels
    _, _, g64 = ro.train_step(c(sc.xyz), c(sc.emb), c(sc.conf), c(sc.dir), c(sc.color), sd, q, c(sc.c2w[:3, 3])[None], c(sc.c2w[:3, :3])[None],
                              c(rays)[None], c(sc.bg_color)[None], c(sc.c2w_nearest)[None], c(sc.c2w_nearest[:, :3, 3])[None],
                              c(sc.intrinsic)[None], c(sc.images_nearest)[None], opt.vsize, c(gt), 1e-3, ro.drop_patch_rays(8, 7, opt.drop_ratio),
                              dtype=torch.float64)
    report = []
    for k, r64 in g64.items():
        r64 = r64.numpy().astype(np.float64)
        if r64.size == 1 or not ("aux_merge_weight_block" in k or k.startswith("neural_points.")):
            continue
        nrm = np.linalg.norm(r64)
        e_gpu = float(np.linalg.norm(got[k].detach().cpu().numpy().astype(np.float64).reshape(r64.shape) - r64) / nrm)
        e_ref = float(np.linalg.norm(gref[k].numpy().astype(np.float64) - r64) / nrm)
        report.append((k, e_gpu, e_ref))
        assert e_gpu <= max(3.0 * e_ref, 2e-4), (k, e_gpu, e_ref)
        if "aux_merge_weight_block" in k:
            # measured: HIP 8e-7 .. 2.1e-5, the reference's fp32 arithmetic 8e-6 .. 1.2e-3 -- the 8e-3 / 3e-3 allowed above against the fp32 oracle is the
            # ORACLE's rounding; against fp64 a 0.3 % systematic error of a merge-weight gradient would fail here by a factor of 30
            assert e_gpu < 1e-4, (k, e_gpu)
    print("C3 full size, relative l2 vs the fp64 graph (HIP | reference arithmetic in fp32): " + "; ".join("%s %.1e | %.1e" % (k.split(".", 1)[1], a_, b_) for k, a_, b_ in report))


def test_c4_scene0101_full_forward_render_and_8_way_shards():
    """BASELINE config C4 (ScanNet scene0101_04 full-res forward render, ray batches sharded across 8 GPUs;
    dev_scripts/w_scannet_etf/scene101_full.sh, run/test_ft.py:139-198) at its full size: 4.0 M points, P 30, max_o 2 M, the whole
    620x460 = 285 200-ray frame through query -> gather/aggregate -> composite in one launch.  (a) the colours of a 2304-ray subset
    against the CPU oracle (C query over all 4 M points + torch aggregate/composite); (b) the 8 scan-line blocks of
    parallel.shard_bounds -- what the 8 ranks render -- reproduce the whole-frame colours BIT FOR BIT."""
    from hybridneuralrendering_amd import scenes
    from hybridneuralrendering_amd.aggregator import PointAggregator
    from hybridneuralrendering_amd.render import HybridRenderer, PointCloud
    from hybridneuralrendering_amd.parallel import shard_bounds
    from oracle import query_oracle as qo, render_oracle as ro
    sc = scenes.make_scene("scene0101", int(4.0e6), 3)
    opt = sc.opt
    assert (opt.P, opt.max_o, opt.SR, opt.K) == (30, 2000000, 24, 8)
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    agg = PointAggregator(opt)
    with torch.no_grad():
        agg.alpha_branch[0].weight.mul_(30.0)
        agg.alpha_branch[0].bias.fill_(30.0)
    sd = {k: v.detach().clone() for k, v in agg.state_dict().items()}
    agg = agg.to(dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cloud = PointCloud(t(sc.xyz), t(sc.emb), t(sc.conf), t(sc.dir), t(sc.color))
    rnd = HybridRenderer(opt, agg, dev)
    pix = scenes.pixel_grid(sc.w, sc.h, 10)
    rays_np = scenes.camera_rays(pix, sc.intrinsic, sc.c2w)
    assert rays_np.shape[0] == 285200
    rays = t(rays_np)
    cam = (t(sc.c2w[:3, 3]), t(sc.c2w[:3, :3]), t(sc.bg_color))
    ref_views = (t(sc.c2w_nearest), t(sc.c2w_nearest[:, :3, 3]), t(sc.intrinsic), t(sc.images_nearest))
    w2c = torch.inverse(torch.from_numpy(sc.c2w_nearest)).to(dev)
    render = lambda r: rnd.render_rays(cloud, r, cam[0], cam[1], cam[2], sc.near, sc.far, ref_views[0], ref_views[1], ref_views[2], ref_views[3],
                                       w2c_nearest=w2c)
    full = render(rays)
    col = full["coarse_raycolor"]
    assert int(full["ray_mask"].sum()) > 0.99 * rays.shape[0] and bool(torch.isfinite(col).all())
    assert float(col.std()) > 0.02                                      # a real image, not a constant
    # (b) the 8 ranks' blocks
    for rank in range(8):
        lo, hi = shard_bounds(rays.shape[0], 8, rank)
        assert hi - lo in (35650,)
        part = render(rays[lo:hi].contiguous())
        assert torch.equal(part["coarse_raycolor"], col[lo:hi]) and torch.equal(part["ray_mask"], full["ray_mask"][lo:hi])
        assert torch.equal(part["coarse_point_opacity"], full["coarse_point_opacity"][lo:hi])
    # (a) oracle on a 48x48-ray block + 1 000 scattered rays
    W = sc.w - 20
    blk = ((200 + np.arange(48))[:, None] * W + (300 + np.arange(48))[None, :]).reshape(-1)
    sel = np.unique(np.concatenate([blk, np.random.default_rng(11).choice(rays_np.shape[0], size=1000, replace=False)]))
    hp = qo.hyperparameters(sc.xyz, opt.vsize, opt.vscale, opt.kernel_size, opt.ranges, opt.radius_limit_scale)
    g = qo.OracleGrid(sc.xyz, hp["origin"], hp["cell"], hp["dims"], opt.query_size, opt.P, opt.max_o)
    q = g.query(sc.c2w[:3, 3], rays_np[sel], qo.tmid_table(sc.near, sc.far, opt.z_depth_dim), opt.SR, opt.K, hp["radius2"], opt.kernel_size)
    c = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    with torch.no_grad():
        oref = ro.render(c(sc.xyz), c(sc.emb), c(sc.conf), c(sc.dir), c(sc.color), sd, q, c(sc.c2w[:3, 3])[None], c(sc.c2w[:3, :3])[None],
                         c(rays_np[sel])[None], c(sc.bg_color)[None], c(sc.c2w_nearest)[None], c(sc.c2w_nearest[:, :3, 3])[None],
                         c(sc.intrinsic)[None], c(sc.images_nearest)[None], opt.vsize)
    exp = oref["full_coarse_raycolor"][0].numpy()
    got = col[torch.from_numpy(sel).to(dev)].cpu().numpy()
    np.testing.assert_array_equal(full["ray_mask"][torch.from_numpy(sel).to(dev)].cpu().numpy(), q["ray_mask"])
    err = float(np.abs(got - exp).max())
    mse = float(np.mean((got.astype(np.float64) - exp.astype(np.float64)) ** 2))
    assert err < 2e-4 and -10 * np.log10(max(mse, 1e-30)) > 80.0, (err, mse)        # fp32 path, summation order differs from the CPU GEMMs
    print("C4 full frame: %d rays, %d valid samples, %d neighbours; max |d colour| vs oracle on %d rays %.2e" % (
        rays.shape[0], int(full["counts"][6]), int(full["counts"][3]), len(sel), err))
