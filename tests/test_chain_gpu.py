"""The fused per-neighbour chain (csrc/chain.hip: hnr_chain_gather + hnr_chain_forward) through the C ABI:
 * every layer's output against an fp64 evaluation of block1 / block3 / alpha_branch + K-sums
   (models/aggregators/point_aggregators.py:948, :957-972, :1005-1026) beside the per-layer fp32-MFMA path on the same rows:
   the two-term fp16 split must stay in the fp32 error class;
 * the same with activations and weights spread over many binary orders of magnitude (per-row / per-layer scaling);
 * tile independence: a sample's result does not depend on which other samples share its 128-row tile."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _world(n_points=60000, w=96, h=72, seed=5, mutate=None):
    from hybridneuralrendering_amd import scenes, _lib
    from hybridneuralrendering_amd import querier as Q
    from hybridneuralrendering_amd.aggregator import PointAggregator
    from hybridneuralrendering_amd.render import HybridRenderer, PointCloud
    from hybridneuralrendering_amd._lib import CNT
    dev = torch.device("cuda:0")
    sc = scenes.make_scene("scene0241", n_points, seed, w=w, h=h)
    opt = sc.opt
    torch.manual_seed(seed)
    agg = PointAggregator(opt)
    with torch.no_grad():
        agg.alpha_branch[0].weight.mul_(30.0)
        agg.alpha_branch[0].bias.fill_(30.0)
        if mutate:
            mutate(agg, sc)
    agg = agg.to(dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cloud = PointCloud(t(sc.xyz), t(sc.emb), t(sc.conf), t(sc.dir), t(sc.color))
    rnd = HybridRenderer(opt, agg, dev)
    pix = scenes.pixel_grid(sc.w, sc.h)
    raydir = t(scenes.camera_rays(pix, sc.intrinsic, sc.c2w))
    campos, camrot = t(sc.c2w[:3, 3]), t(sc.c2w[:3, :3]).contiguous()
    grid, hp = rnd.querier._grid_for(cloud.xyz[None])
    tmid = rnd.querier._tmid_for(sc.near, sc.far, opt.z_depth_dim, raydir.shape[0], dev)
    q = Q.march_query(grid, campos, raydir, tmid, opt.SR, opt.K, np.float32(hp[0] ** 2), opt.kernel_size, pad=False)
    c = q["counts"].cpu()
    n_valid, n_rows = int(c[CNT["SAMPLES_VALID"]]), int(c[CNT["NEIGHBOURS"]])
    assert n_valid > 500
    L, p = _lib.lib(), _lib.ptr
    R, SR, K = q["sample_pidx"].shape
    i32 = lambda n: torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
    vs_item, vs_off, vs_cnt = i32(n_valid), i32(n_valid), i32(n_valid)
    scratch, overflow = i32(2 * ((R * SR + 1023) // 1024) + 2), torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(L.hnr_sample_plan(p(q["work"]), p(q["sample_pidx"]), p(q["counts"]), K, R * SR, p(vs_item), p(vs_off), p(vs_cnt), n_valid, n_rows,
                                 p(scratch), p(overflow), _lib.stream()), "hnr_sample_plan")
    return dict(sc=sc, opt=opt, agg=agg, cloud=cloud, rnd=rnd, raydir=raydir, campos=campos, camrot=camrot, q=q, n_valid=n_valid,
                n_rows=n_rows, vs_item=vs_item, vs_off=vs_off, vs_cnt=vs_cnt, dev=dev, R=R, SR=SR, K=K)


def _per_layer_inputs(W):
    """The per-layer path's row buffers (hnr_gather_rows, SPLIT layout): Xd [rows,64], extras [rows,7], wagg, row_pid."""
    from hybridneuralrendering_amd import _lib
    L, p, dev, cloud, q = _lib.lib(), _lib.ptr, W["dev"], W["cloud"], W["q"]
    n_rows, n_valid = W["n_rows"], W["n_valid"]
    Xd = torch.zeros((n_rows, 64), dtype=torch.float32, device=dev)
    C = torch.zeros((n_rows, 264), dtype=torch.float32, device=dev)
    wagg = torch.empty((n_rows,), dtype=torch.float32, device=dev)
    row_pid = torch.empty((n_rows,), dtype=torch.int32, device=dev)
    _lib.check(L.hnr_gather_rows(p(cloud.xyz), p(cloud.emb), p(cloud.conf), p(cloud.dir), p(cloud.color), cloud.F, p(q["sample_pidx"]),
                                 p(q["sample_loc_w"]), p(W["raydir"]), p(W["campos"]), p(W["camrot"]), p(W["vs_item"]), p(W["vs_off"]),
                                 p(W["vs_cnt"]), p(q["counts"]), W["SR"], W["K"], n_valid, p(Xd), 64, p(C), 264, p(wagg), None, None,
                                 p(row_pid), _lib.stream()), "hnr_gather_rows")
    return Xd, C[:, 256:263].clone(), wagg, row_pid, C


def _fp64_chain(W, Xd, ext, wagg, row_pid, ptab):
    agg = W["agg"]
    d = lambda t: t.detach().double()
    sl = float(agg.block1[1].negative_slope)
    lk = lambda x: torch.where(x > 0, x, x * sl)
    H1 = lk(d(Xd[:, :60]) @ d(agg.block1[0].weight[:, 224:284]).T + d(agg.block1[0].bias) + d(ptab)[row_pid.long()])
    H2 = lk(H1 @ d(agg.block1[2].weight).T + d(agg.block1[2].bias))
    H3 = lk(torch.cat([H2, d(ext)], dim=1) @ d(agg.block3[0].weight).T + d(agg.block3[0].bias))
    H4 = lk(H3 @ d(agg.block3[2].weight).T + d(agg.block3[2].bias))
    alpha = H4 @ d(agg.alpha_branch[0].weight).reshape(-1) + d(agg.alpha_branch[0].bias)
    sp = torch.nn.functional.softplus(alpha - 1.0)
    seg = torch.repeat_interleave(torch.arange(W["n_valid"], device=W["dev"]), W["vs_cnt"].long())
    X5 = torch.zeros((W["n_valid"], 256), dtype=torch.float64, device=W["dev"]).index_add_(0, seg, H4 * d(wagg)[:, None])
    sig = torch.zeros((W["n_valid"],), dtype=torch.float64, device=W["dev"]).index_add_(0, seg, sp * d(wagg))
    return [H1, H2, H3, H4], X5, sig


def _run_chain(W, dbg_layer=None, cap=None):
    from hybridneuralrendering_amd import _lib
    L, p, dev, cloud, q, rnd = _lib.lib(), _lib.ptr, W["dev"], W["cloud"], W["q"], W["rnd"]
    n_valid = W["n_valid"] if cap is None else cap
    ws = torch.empty((int(L.hnr_chain_workspace_bytes(n_valid)),), dtype=torch.uint8, device=dev)
    X5 = torch.full((n_valid, 280), float("nan"), dtype=torch.float32, device=dev)
    sigma = torch.full((n_valid,), float("nan"), dtype=torch.float32, device=dev)
    ptab = rnd.point_table(cloud)
    _lib.check(L.hnr_chain_gather(p(cloud.xyz), p(cloud.conf), p(cloud.dir), p(cloud.color), p(q["sample_pidx"]), p(q["sample_loc_w"]),
                                  p(W["raydir"]), p(W["campos"]), p(W["camrot"]), p(W["vs_item"]), p(q["counts"]), W["SR"], W["K"], n_valid,
                                  p(ws), p(X5), 280, None, None, _lib.stream()), "hnr_chain_gather")
    dbg = None
    if dbg_layer is not None:
        dbg = torch.zeros((((n_valid + 15) // 16) * 128, 256), dtype=torch.float32, device=dev)
    _lib.check(L.hnr_chain_forward(p(ws), p(ptab), int(ptab.stride(0)), p(W["agg"].packed_chain()), p(q["counts"]), n_valid,
                                   float(W["agg"].block1[1].negative_slope), p(X5), 280, p(sigma), p(dbg) if dbg is not None else None,
                                   dbg_layer or 0, _lib.stream()), "hnr_chain_forward")
    torch.cuda.synchronize()
    return X5, sigma, dbg, ptab


def _padded_rows(W):
    """packed row index -> row of the padded layout (8 slots per valid sample)"""
    seg = torch.repeat_interleave(torch.arange(W["n_valid"], device=W["dev"]), W["vs_cnt"].long())
    k = torch.arange(W["n_rows"], device=W["dev"]) - W["vs_off"].long()[seg]
    return seg * 8 + k


def _per_layer_f32(W, Xd, Cbuf, wagg, row_pid, ptab):
    """The per-layer path on fp32 MFMA (hnr_linear_f32*), as HybridRenderer(dense='f32') runs it."""
    from hybridneuralrendering_amd import _lib
    L, p = _lib.lib(), _lib.ptr
    pk = W["agg"].packed()
    sl = pk["slope"]
    n_rows = W["n_rows"]
    dev = W["dev"]
    B = torch.empty((n_rows, 256), dtype=torch.float32, device=dev)
    A = torch.empty((n_rows, 256), dtype=torch.float32, device=dev)
    pk["b1_dist"].gather_add(Xd, ptab, row_pid, out=B, act=True, slope=sl, K=60)
    H1 = B.clone()
    pk["b1"][1](B, out=Cbuf, act=True, slope=sl)
    H2 = Cbuf[:, :256].clone()
    pk["b3"][0](Cbuf, out=A, act=True, slope=sl, K=263)
    H3 = A.clone()
    pk["b3"][1](A, out=B, act=True, slope=sl)
    X5 = torch.empty((W["n_valid"], 280), dtype=torch.float32, device=dev)
    sigma = torch.empty((W["n_valid"],), dtype=torch.float32, device=dev)
    _lib.check(L.hnr_ksum(p(B), 256, p(wagg), p(pk["alpha_w"]), p(pk["alpha_b"]), p(W["vs_item"]), p(W["vs_off"]), p(W["vs_cnt"]),
                          p(W["raydir"]), p(W["q"]["counts"]), W["SR"], W["n_valid"], p(X5), 280, p(sigma), _lib.stream()), "hnr_ksum")
    return [H1, H2, H3, B], X5, sigma


def _rel(a, ref):
    """max |a - ref| relative to the largest |ref| of the same row (a dot-product error scales with sum |a w|, not with the entry)"""
    a, ref = a.double(), ref.double()
    den = ref.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30) if ref.dim() > 1 else ref.abs().clamp_min(1e-30)
    return float(((a - ref).abs() / den).max())


def _check_against_fp64(W, factor=2.5, floor=3e-7):
    Xd, ext, wagg, row_pid, Cbuf = _per_layer_inputs(W)
    prow = _padded_rows(W)
    X5c, sigc, _, ptab = _run_chain(W)
    Hs64, X564, sig64 = _fp64_chain(W, Xd, ext, wagg, row_pid, ptab)
    Hs32, X532, sig32 = _per_layer_f32(W, Xd, Cbuf, wagg, row_pid, ptab)
    report = []
    for layer in range(4):
        _, _, dbg, _ = _run_chain(W, dbg_layer=layer)
        got = dbg[prow]
        e_chain, e_f32 = _rel(got, Hs64[layer]), _rel(Hs32[layer], Hs64[layer])
        report.append((layer, e_chain, e_f32))
        assert e_chain <= factor * e_f32 + floor, "layer %d: fused chain %.3e vs fp32-MFMA per-layer path %.3e (both against fp64)" % (layer, e_chain, e_f32)
    e5c, e5f = _rel(X5c[:, :256], X564), _rel(X532[:, :256], X564)
    esc, esf = _rel(sigc, sig64), _rel(sig32, sig64)
    report.append(("X5", e5c, e5f)); report.append(("sigma", esc, esf))
    assert e5c <= factor * e5f + floor and esc <= factor * esf + 1e-6, report
    # the view-direction encoding columns are the per-layer path's, bit for bit
    assert torch.equal(X5c[:, 256:280], X532[:, 256:280])
    return report


def test_chain_layers_are_fp32_class_against_fp64(capsys):
    W = _world()
    rep = _check_against_fp64(W)
    with capsys.disabled():
        print("\n[chain vs fp64] (stage, fused f16x2 chain, per-layer fp32 MFMA): " + "; ".join("%s %.2e %.2e" % r for r in rep))


def test_chain_wide_dynamic_range(capsys):
    """Rows and layers of very different magnitude (per-point embedding scales 2^-6 .. 2^6, a layer scaled by 3e-4 and its successor by
    3e+3) and hidden units spread over 2^12 inside a row: the per-row activation scales and per-layer weight scales keep the fp16
    split in range -- no overflow, and 22 significant bits for every element within 2^18 of its row's (layer's) maximum, which is
    the documented reach of the scheme (csrc/chain.hip: below that the error floor is 2^-40 of the maximum)."""
    def mutate(agg, sc):
        g = torch.Generator().manual_seed(3)
        s1 = torch.exp2(torch.randint(-6, 7, (256,), generator=g).float())
        agg.block1[2].weight.mul_(s1[:, None]); agg.block1[2].bias.mul_(s1)
        agg.block3[0].weight[:, :256].div_(s1[None, :])
        agg.block3[2].weight.mul_(3e-4); agg.alpha_branch[0].weight.mul_(1.0 / 3e-4)
        sc.emb *= np.exp2(np.random.default_rng(1).integers(-6, 7, size=(sc.emb.shape[0], 1))).astype(np.float32)
    W = _world(seed=6, mutate=mutate)
    rep = _check_against_fp64(W, factor=3.0, floor=1e-6)
    with capsys.disabled():
        print("\n[chain wide range] " + "; ".join("%s %.2e %.2e" % r for r in rep))


def test_chain_result_is_independent_of_tile_composition():
    """Per-ROW scaling: rendering a subset of the samples (other tile mates) reproduces their sums bit for bit, which is what makes
    chunked / ray-sharded renders equal the whole-frame render exactly."""
    from hybridneuralrendering_amd import _lib
    W = _world(seed=7)
    X5a, siga, _, _ = _run_chain(W)
    # drop the first 5 valid samples: every remaining sample moves to another tile slot
    W2 = dict(W)
    W2["vs_item"] = W["vs_item"][5:].contiguous()
    cnt = W["q"]["counts"].clone()
    cnt[_lib.CNT["SAMPLES_VALID"]] -= 5
    W2["q"] = dict(W["q"], counts=cnt)
    W2["n_valid"] = W["n_valid"] - 5
    X5b, sigb, _, _ = _run_chain(W2)
    assert torch.equal(X5a[5:], X5b) and torch.equal(siga[5:], sigb)


def test_chain_sample_classes_are_bit_identical():
    """hnr_chain_plan(classes = 1 / 2) lists the samples with few neighbours after the others (1..4 neighbours: 4 row slots; with
    classes = 2: 3..4 neighbours 4 slots, 1..2 neighbours 2 slots) and the gather / chain kernels size their row slots by class: every
    sample's sums are bit for bit those of the one-class layout (an empty slot adds an exact zero)."""
    from hybridneuralrendering_amd import _lib
    L, p = _lib.lib(), _lib.ptr
    CNT = _lib.CNT
    W = _world(seed=7)
    q, dev, K = W["q"], W["dev"], W["K"]
    n_valid, n_items = W["n_valid"], W["R"] * W["SR"]
    scratch = torch.empty((3 * ((n_items + 1023) // 1024) + 3,), dtype=torch.int32, device=dev)
    def plan(classes, cap=n_valid):
        cnt = q["counts"].clone()
        vs = torch.full((n_valid,), -1, dtype=torch.int32, device=dev)
        _lib.check(L.hnr_chain_plan(p(q["work"]), p(q["sample_pidx"]), p(cnt), K, n_items, classes, p(vs), cap, p(scratch), _lib.stream()),
                   "hnr_chain_plan")
        torch.cuda.synchronize()
        return vs, cnt
    vs0, cnt0 = plan(0)
    assert torch.equal(vs0, W["vs_item"]) and int(cnt0[CNT["SAMPLES_SMALL"]]) == 0 and int(cnt0[CNT["SAMPLES_TINY"]]) == 0
    X5a, siga, _, _ = _run_chain(W)
    have = int(L.hnr_chain_classes())
    assert 0 <= have <= 2
    if have < 2:
        assert L.hnr_chain_plan(p(q["work"]), p(q["sample_pidx"]), p(cnt0), K, n_items, have + 1, p(vs0), n_valid, p(scratch), _lib.stream()) != 0
    if have == 0:
        pytest.skip("the selected chain kernel has one sample class")
    nb_all = (q["sample_pidx"].reshape(-1, K) >= 0).sum(-1)
    for classes in range(1, have + 1):
        vs1, cnt1 = plan(classes)
        n_small, n_tiny = int(cnt1[CNT["SAMPLES_SMALL"]]), int(cnt1[CNT["SAMPLES_TINY"]])
        n_big = n_valid - n_small - n_tiny
        assert 0 < n_small < n_valid and (n_tiny > 0) == (classes == 2)
        nb = nb_all[vs1.long()]
        lo_small = 1 if classes == 1 else 3
        assert bool((nb[:n_big] > 4).all()) and bool(((nb[n_big:n_big + n_small] >= lo_small) & (nb[n_big:n_big + n_small] <= 4)).all())
        assert bool(((nb[n_big + n_small:] >= 1) & (nb[n_big + n_small:] <= 2)).all())
        for lo, hi in ((0, n_big), (n_big, n_big + n_small), (n_big + n_small, n_valid)):       # (ray, slot) order inside every class
            assert bool((vs1[lo + 1:hi] > vs1[lo:hi - 1]).all())
        pos = torch.searchsorted(W["vs_item"], vs1)                 # where each sample sits in the one-class list
        assert torch.equal(W["vs_item"][pos], vs1)
        X5b, sigb, _, _ = _run_chain(dict(W, vs_item=vs1, q=dict(q, counts=cnt1)))
        assert torch.equal(X5a[pos], X5b) and torch.equal(siga[pos], sigb)
        # capacities that cut the list inside the first class, between the classes and inside the last one: the later classes are the ones dropped
        for cap in (n_big - 3, n_big + max(1, n_small // 2), n_valid - max(1, (n_tiny or n_small) // 3)):
            vs, cnt = plan(classes, cap)
            first = min(n_big, cap)
            second = min(n_small, cap - first)
            assert int(cnt[CNT["SAMPLES_SMALL"]]) == second and int(cnt[CNT["SAMPLES_TINY"]]) == cap - first - second
            assert torch.equal(vs[:cap], vs1[:cap]) and bool((vs[cap:] == -1).all())
            X5c, sigc, _, _ = _run_chain(dict(W, vs_item=vs, q=dict(q, counts=cnt)), cap=cap)
            assert torch.equal(X5a[pos[:cap]], X5c) and torch.equal(siga[pos[:cap]], sigc)


def test_point_records_gather_is_bit_identical():
    """hnr_point_records interleaves xyz / conf / dir / colour into 48-byte records; hnr_chain_gather_rec reading them writes the same
    workspace, X5 view-direction columns and weight / confidence outputs as hnr_chain_gather reading the four buffers."""
    from hybridneuralrendering_amd import _lib
    from hybridneuralrendering_amd.render import HnrError
    L, p = _lib.lib(), _lib.ptr
    W = _world(seed=11)
    c, q, dev, nv = W["cloud"], W["q"], W["dev"], W["n_valid"]
    n = c.xyz.shape[0]
    rec = torch.full((n, 12), float("nan"), dtype=torch.float32, device=dev)
    _lib.check(L.hnr_point_records(p(c.xyz), p(c.conf), p(c.dir), p(c.color), n, p(rec), _lib.stream()), "hnr_point_records")
    torch.cuda.synchronize()
    want = torch.cat([c.xyz.reshape(n, 3), c.conf.reshape(n, 1), c.dir.reshape(n, 3), c.color.reshape(n, 3), torch.zeros((n, 2), device=dev)], dim=1)
    assert torch.equal(rec, want)
    R, SR, K = q["sample_pidx"].shape
    outs = []
    for use_rec in (False, True):
        ws = torch.zeros((int(L.hnr_chain_workspace_bytes(nv)),), dtype=torch.uint8, device=dev)
        X5 = torch.zeros((nv, 280), dtype=torch.float32, device=dev)
        wo, co = torch.zeros((R, SR, K), dtype=torch.float32, device=dev), torch.zeros((R, SR, K), dtype=torch.float32, device=dev)
        tail = (p(q["sample_pidx"]), p(q["sample_loc_w"]), p(W["raydir"]), p(W["campos"]), p(W["camrot"]), p(W["vs_item"]), p(q["counts"]), SR, K, nv,
                p(ws), p(X5), 280, p(wo), p(co), _lib.stream())
        if use_rec:
            _lib.check(L.hnr_chain_gather_rec(p(rec), *tail), "hnr_chain_gather_rec")
        else:
            _lib.check(L.hnr_chain_gather(p(c.xyz), p(c.conf), p(c.dir), p(c.color), *tail), "hnr_chain_gather")
        torch.cuda.synchronize()
        outs.append((ws, X5, wo, co))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert float(outs[0][2].abs().sum()) > 0
    with pytest.raises(HnrError):
        _lib.check(L.hnr_chain_gather_rec(None, *tail), "hnr_chain_gather_rec")
    with pytest.raises(HnrError):
        _lib.check(L.hnr_point_records(None, None, None, None, 5, None, None), "hnr_point_records")


def test_chain_capacity_bounds_and_bad_arguments():
    from hybridneuralrendering_amd import _lib
    from hybridneuralrendering_amd._lib import HnrError
    W = _world(seed=8, n_points=30000, w=64, h=48)
    full, sig_full, _, _ = _run_chain(W)
    cap = W["n_valid"] - 37                                  # a capacity below the device-side count: only `cap` samples are produced
    part, sig_part, _, _ = _run_chain(W, cap=cap)
    assert torch.equal(part, full[:cap]) and torch.equal(sig_part, sig_full[:cap])
    L, p = _lib.lib(), _lib.ptr
    with pytest.raises(HnrError):
        _lib.check(L.hnr_chain_gather(None, None, None, None, None, None, None, None, None, None, None, 24, 9, 16, None, None, 280, None, None,
                                      None), "hnr_chain_gather")
    with pytest.raises(HnrError):
        _lib.check(L.hnr_chain_forward(None, None, 256, None, None, 16, ctypes.c_float(1.5), None, 280, None, None, 0, None), "hnr_chain_forward")
    # whole 16-sample blocks, plus two: each of the three slot classes may end in a partial block
    assert L.hnr_chain_workspace_bytes(0) == 8 * (8192 + 1280 + 128) and L.hnr_chain_workspace_bytes(17) == 16 * (8192 + 1280 + 128)
