"""GPU parity: HIP grid + march/k-NN (through the C ABI) vs the C oracle -- bit-exact (index work)."""
import numpy as np
import pytest
import torch

from hybridneuralrendering_amd import scenes

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _case(seed, n, P, max_o, K, SR, R, D=200, size=(1.0, 0.8, 0.5), near=0.05, far=1.5, wh=(64, 48)):
    from oracle import query_oracle as qo
    rng = np.random.default_rng(seed)
    xyz, _ = scenes.room_cloud(n, seed, size=size, n_clutter=3, thickness=0.003)
    hp = qo.hyperparameters(xyz, [0.008] * 3, [2, 2, 2], [3, 3, 3], [-10.0] * 3 + [10.0] * 3, 4.0)
    s = np.asarray(size)
    cam = scenes.look_at(-0.3 * s + [0, 0, 0.25 * s[2]], 0.4 * s * [1, 1, -0.5])
    K3 = scenes.pinhole(wh[0], wh[1], 0.9 * wh[0])
    pix = scenes.pixel_grid(wh[0], wh[1])
    sel = np.sort(rng.choice(pix.shape[0], size=min(R, pix.shape[0]), replace=False))
    rays = scenes.camera_rays(pix[sel], K3, cam)
    return dict(xyz=xyz, hp=hp, campos=cam[:3, 3].copy(), camrot=cam[:3, :3].copy(), rays=rays,
                tmid=qo.tmid_table(near, far, D), P=P, max_o=max_o, K=K, SR=SR)


def _run_both(cs, tmid=None):
    from oracle import query_oracle as qo
    from hybridneuralrendering_amd import querier as Q
    hp = cs["hp"]
    tm = cs["tmid"] if tmid is None else tmid
    og = qo.OracleGrid(cs["xyz"], hp["origin"], hp["cell"], hp["dims"], [3, 3, 3], cs["P"], cs["max_o"])
    ref = og.query(cs["campos"], cs["rays"], tm, cs["SR"], cs["K"], hp["radius2"], [3, 3, 3], want_full=True)
    d = _dev()
    xyz = torch.from_numpy(cs["xyz"]).to(d)
    g = Q.VoxelGrid(xyz, hp["origin"], hp["cell"], hp["dims"], [3, 3, 3], cs["P"], cs["max_o"])
    res = Q.march_query(g, torch.from_numpy(cs["campos"]).to(d), torch.from_numpy(cs["rays"]).to(d),
                        torch.from_numpy(np.ascontiguousarray(tm)).to(d), cs["SR"], cs["K"], hp["radius2"], [3, 3, 3])
    return og, ref, g, res


def _assert_query_equal(ref, res, cs):
    np.testing.assert_array_equal(res["ray_nsamp"].cpu().numpy(), ref["full_nsamp"])
    np.testing.assert_array_equal(res["sample_loc_w"].cpu().numpy(), ref["full_loc"])
    np.testing.assert_array_equal(res["sample_pidx"].cpu().numpy(), ref["full_pidx"])
    np.testing.assert_array_equal(res["ray_mask"].cpu().numpy(), ref["ray_mask"])
    c = res["counts"].cpu().numpy()
    from hybridneuralrendering_amd._lib import CNT
    rc = ref["counts"]
    assert c[CNT["RAYS_HIT"]] == rc["n_hit_rays"]
    assert c[CNT["SAMPLES"]] == rc["n_samples"]
    assert c[CNT["NEIGHBOURS"]] == rc["n_neighbours"]
    assert c[CNT["CELLS_VISITED"]] == rc["n_cells_visited"]
    assert c[CNT["CANDIDATES"]] == rc["n_candidates"]
    assert c[CNT["SAMPLES_VALID"]] == rc["n_valid_samples"]


@pytest.mark.parametrize("seed,P,max_o", [(0, 4, 100000), (1, 26, 100000), (3, 4, 700), (4, 1, 300)])
def test_grid_tables_match_oracle(seed, P, max_o):
    cs = _case(seed, 8000, P, max_o, 8, 6, 64)
    og, ref, g, res = _run_both(cs)
    occ, c2o, o2p, onp = og.tables()
    d_occ, d_cnt, d_first = (t.cpu().numpy() for t in g.export_dense())
    np.testing.assert_array_equal(d_occ, occ)
    exp_cnt = np.where(c2o >= 0, np.minimum(P, onp[np.maximum(c2o, 0)]), -1)
    np.testing.assert_array_equal(d_cnt, exp_cnt)
    exp_first = np.where((c2o >= 0) & (exp_cnt > 0), o2p[np.maximum(c2o, 0), 0], -1)
    np.testing.assert_array_equal(d_first, exp_first)
    info = og.info()
    assert g.stats["n_occ"] == info["n_occ"]
    assert g.stats["n_inbounds"] == info["n_inbounds"]
    assert g.stats["n_dropped_voxels"] == info["n_dropped_voxels"]
    assert g.stats["n_cells_over_P"] == info["n_cells_over_P"]
    assert g.stats["n_dilated"] == info["n_dilated"]
    if max_o < 1000:
        assert info["n_dropped_voxels"] > 0


@pytest.mark.parametrize("seed,n,P,max_o,K,SR,R", [
    (0, 8000, 4, 100000, 8, 6, 512),
    (1, 60000, 26, 100000, 8, 24, 1024),     # dense: early exit after layer 0 happens
    (2, 8000, 3, 100000, 4, 5, 300),
    (3, 8000, 4, 700, 8, 6, 300),            # max_o overflow
    (5, 8000, 8, 100000, 1, 3, 300),
    (6, 20000, 12, 100000, 16, 80, 700),     # SR > 64 (synthetic scenes use 80), K = 16
    (7, 20000, 9, 100000, 5, 7, 257),        # odd K / odd SR*K: scalar store path
    (9, 20000, 12, 100000, 9, 6, 300),       # any K <= HNR_MAX_K is built (the reference compiles `#define KN <K>` for any value, :110)
    (10, 20000, 30, 100000, 27, 4, 200),
])
def test_march_query_bit_exact(seed, n, P, max_o, K, SR, R):
    cs = _case(seed, n, P, max_o, K, SR, R)
    og, ref, g, res = _run_both(cs)
    _assert_query_equal(ref, res, cs)
    assert ref["counts"]["n_valid_rays"] > 0


def test_per_ray_depth_tables():
    cs = _case(8, 20000, 8, 100000, 8, 12, 400)
    rng = np.random.default_rng(0)
    D = cs["tmid"].shape[0]
    jit = (cs["tmid"][None, :] + rng.uniform(-0.002, 0.002, size=(cs["rays"].shape[0], D))).astype(np.float32)
    jit = np.sort(jit, axis=1)
    og, ref, g, res = _run_both(cs, tmid=jit)
    _assert_query_equal(ref, res, cs)


def test_compaction_and_pers_match_reference_ops():
    from hybridneuralrendering_amd import querier as Q
    cs = _case(9, 30000, 26, 100000, 8, 24, 900, far=0.45)     # short far plane: part of the rays reach nothing
    og, ref, g, res = _run_both(cs)
    d = _dev()
    rays = torch.from_numpy(cs["rays"]).to(d)
    campos = torch.from_numpy(cs["campos"]).to(d)
    camrot = torch.from_numpy(cs["camrot"]).to(d)
    pidx, pers, loc_w, dirs, row = Q.compact_rays(res, rays, campos, camrot)
    np.testing.assert_array_equal(pidx.cpu().numpy(), ref["sample_pidx"])
    np.testing.assert_array_equal(loc_w.cpu().numpy(), ref["sample_loc_w"])
    rows = np.nonzero(ref["ray_mask"])[0]
    assert 0 < len(rows) < len(ref["ray_mask"])          # some rays are dropped: the compaction does work
    exp_dirs = np.broadcast_to(cs["rays"][rows][:, None, :], loc_w.shape)
    np.testing.assert_array_equal(dirs.cpu().numpy(), exp_dirs)
    # w2pers with the reference's torch ops on CPU (query_point_indices_worldcoords.py:96-103)
    lw = torch.from_numpy(ref["sample_loc_w"])[None]
    cp = torch.from_numpy(cs["campos"])[None]
    cr = torch.from_numpy(cs["camrot"])[None]
    shift = lw - cp[:, None, :]
    xyz_c = torch.sum(shift[..., None, :] * torch.transpose(cr, 1, 2)[:, None, None, ...], dim=-1)
    exp = torch.stack([xyz_c[..., 0] / xyz_c[..., 2], xyz_c[..., 1] / xyz_c[..., 2], xyz_c[..., 2]], dim=-1)[0]
    np.testing.assert_allclose(pers.cpu().numpy(), exp.numpy(), rtol=2e-6, atol=1e-6)
    exp_row = np.full(len(ref["ray_mask"]), -1, np.int32)
    exp_row[rows] = np.arange(len(rows))
    np.testing.assert_array_equal(row.cpu().numpy(), exp_row)


def test_dropin_querier_tuple():
    """lighting_fast_querier.query_points -> the reference's 7-tuple (:93)."""
    from oracle import query_oracle as qo
    from hybridneuralrendering_amd.querier import lighting_fast_querier
    sc = scenes.make_scene("scene0241", 120000, 2, w=96, h=72)
    opt = sc.opt
    d = _dev()
    q = lighting_fast_querier(d, opt)
    pix = scenes.pixel_grid(sc.w, sc.h)
    rays = scenes.camera_rays(pix, sc.intrinsic, sc.c2w)
    xyz = torch.from_numpy(sc.xyz).to(d)
    out = q.query_points(torch.from_numpy(pix)[None].to(d), None, xyz[None], None, sc.h, sc.w, sc.intrinsic,
                         sc.near, sc.far, torch.from_numpy(rays)[None].to(d),
                         torch.from_numpy(sc.c2w[:3, 3].copy())[None].to(d), torch.from_numpy(sc.c2w[:3, :3].copy())[None].to(d))
    pidx, loc_pers, loc_w, dirs, ray_mask, vsize, ranges_np = out
    hp = qo.hyperparameters(sc.xyz, opt.vsize, opt.vscale, opt.kernel_size, opt.ranges, opt.radius_limit_scale)
    np.testing.assert_array_equal(ranges_np, hp["ranges_np"])
    og = qo.OracleGrid(sc.xyz, hp["origin"], hp["cell"], hp["dims"], opt.query_size, opt.P, opt.max_o)
    ref = og.query(sc.c2w[:3, 3], rays, qo.tmid_table(sc.near, sc.far, opt.z_depth_dim), opt.SR, opt.K, hp["radius2"], opt.kernel_size)
    assert pidx.shape == (1,) + ref["sample_pidx"].shape and pidx.dtype == torch.int32
    np.testing.assert_array_equal(pidx[0].cpu().numpy(), ref["sample_pidx"])
    np.testing.assert_array_equal(loc_w[0].cpu().numpy(), ref["sample_loc_w"])
    assert ray_mask.dtype == torch.int8 and ray_mask.shape == (1, rays.shape[0])
    np.testing.assert_array_equal(ray_mask[0].cpu().numpy(), ref["ray_mask"])
    assert dirs.shape == loc_w.shape and loc_pers.shape == loc_w.shape
    assert vsize is opt.vsize
    assert ref["counts"]["n_valid_rays"] > 1000
    # second call reuses the grid (same cloud version)
    g0 = q._grid
    q.query_points(torch.from_numpy(pix)[None].to(d), None, xyz[None], None, sc.h, sc.w, sc.intrinsic, sc.near, sc.far,
                   torch.from_numpy(rays)[None].to(d), torch.from_numpy(sc.c2w[:3, 3].copy())[None].to(d),
                   torch.from_numpy(sc.c2w[:3, :3].copy())[None].to(d))
    assert q._grid is g0
    q.clean_up()


def test_edge_cases():
    from hybridneuralrendering_amd import querier as Q
    from hybridneuralrendering_amd._lib import HnrError
    cs = _case(10, 4000, 4, 100000, 8, 6, 64)
    hp = cs["hp"]
    d = _dev()
    xyz = torch.from_numpy(cs["xyz"]).to(d)
    g = Q.VoxelGrid(xyz, hp["origin"], hp["cell"], hp["dims"], [3, 3, 3], 4, 100000)
    campos = torch.from_numpy(cs["campos"]).to(d)
    tm = torch.from_numpy(cs["tmid"]).to(d)
    # zero rays
    res = Q.march_query(g, campos, torch.zeros((0, 3), device=d), tm, 6, 8, hp["radius2"], [3, 3, 3])
    assert res["sample_pidx"].shape == (0, 6, 8)
    out = Q.compact_rays(res, torch.zeros((0, 3), device=d), campos, torch.eye(3, device=d))
    assert out[0].shape == (0, 6, 8)
    # rays that miss everything
    away = torch.tensor([[0.0, 0.0, 1.0], [0.0, 0.0, 1.0]], device=d) * 1.0
    res = Q.march_query(g, torch.tensor([50.0, 50.0, 50.0], device=d), away, tm, 6, 8, hp["radius2"], [3, 3, 3])
    assert int(res["ray_mask"].sum()) == 0 and int(res["counts"][1]) == 0
    assert torch.all(res["sample_pidx"] == -1) and torch.all(res["sample_loc_w"] == 0)
    # NaN ray direction: treated as out of bounds, no crash
    bad = torch.tensor([[float("nan"), 0.0, 1.0]], device=d)
    res = Q.march_query(g, campos, bad, tm, 6, 8, hp["radius2"], [3, 3, 3])
    assert int(res["ray_nsamp"][0]) == 0
    # bad arguments are reported, not crashed on
    with pytest.raises(HnrError):
        Q.march_query(g, campos, away, tm, 6, 33, hp["radius2"], [3, 3, 3])     # K > HNR_MAX_K = 32
    with pytest.raises(HnrError):
        Q.VoxelGrid(xyz, hp["origin"], hp["cell"], [0, 4, 4], [3, 3, 3], 4, 10)
    with pytest.raises(HnrError):
        Q.VoxelGrid(xyz.cpu(), hp["origin"], hp["cell"], hp["dims"], [3, 3, 3], 4, 10)
    mn, mx = Q.points_bounds(xyz)
    np.testing.assert_array_equal(mn, cs["xyz"].min(0))
    np.testing.assert_array_equal(mx, cs["xyz"].max(0))


def test_mid_size_scene():
    """scene0241-like room, 400k points, 20k rays: bit-exact and with the counters the roofline uses."""
    cs = _case(11, 400000, 26, 610000, 8, 24, 20000, D=400, size=(4.0, 3.0, 2.0), near=0.1, far=8.0, wh=(200, 150))
    og, ref, g, res = _run_both(cs)
    _assert_query_equal(ref, res, cs)
    assert ref["counts"]["n_samples"] > 50000


@pytest.mark.parametrize("seed,n,P,R", [(0, 8000, 4, 400), (1, 60000, 26, 1500), (5, 60000, 12, 1500)])
def test_sorted_neighbour_order_is_the_reference_set_in_ascending_distance(seed, n, P, R):
    """hnr_query_params.knn_order = 1 (the production option for order-free consumers): per kept sample the same neighbour SET as the
    reference's insertion rule (/root/reference/models/neural_points/query_point_indices_worldcoords.py:494-513, checked against the C
    oracle), listed by ascending d^2 -- valid ids first, -1 after them -- and every other output of the query unchanged."""
    from hybridneuralrendering_amd import querier as Q
    cs = _case(seed, n, P, 100000, 8, 12, R)
    og, ref, g, res = _run_both(cs)
    d = _dev()
    srt = Q.march_query(g, torch.from_numpy(cs["campos"]).to(d), torch.from_numpy(cs["rays"]).to(d), torch.from_numpy(np.ascontiguousarray(cs["tmid"])).to(d),
                        cs["SR"], cs["K"], cs["hp"]["radius2"], [3, 3, 3], knn_order=1)
    for k in ("ray_nsamp", "sample_loc_w", "ray_mask", "counts"):
        assert torch.equal(srt[k], res[k]), k
    a, b = srt["sample_pidx"].cpu().numpy().reshape(-1, 8), ref["full_pidx"].reshape(-1, 8)
    loc = ref["full_loc"].reshape(-1, 3)
    np.testing.assert_array_equal(np.sort(a, axis=1), np.sort(b, axis=1))                 # the same set, slot by slot after sorting
    full = int((np.sort(b, axis=1)[:, 0] >= 0).sum())
    assert full > 100 and (b >= 0).any(axis=1).sum() > full                                # both full and partly filled samples occur
    valid = a >= 0
    assert (valid[:, :-1] >= valid[:, 1:]).all()                                          # valid ids are a prefix
    x = cs["xyz"][np.maximum(a, 0)]                                                        # [S, 8, 3]
    dv = x - loc[:, None, :]
    d2 = (dv[..., 0] * dv[..., 0] + dv[..., 1] * dv[..., 1]) + dv[..., 2] * dv[..., 2]     # float32, the kernel's operation order
    d2 = np.where(valid, d2, np.float32(np.inf))
    assert (d2[:, :-1] <= d2[:, 1:]).all()
    assert (a != b).any()                                                                  # and it is not the reference's slot order


def test_sorted_neighbour_order_with_exact_distance_ties(capsys):
    """Exact d^2 ties (a cloud mirrored in z, camera and rays in the plane z = 0: every point and its mirror image are equidistant from every
    sample, bit for bit).  knn_order = 0 replays the reference's insertion history slot for slot, ties included (first maximum evicted,
    strict `<`: query_point_indices_worldcoords.py:493-513).  knn_order = 1 keeps the K smallest (d^2, enumeration order) keys: where two
    equidistant points compete for the LAST place of a full list it may keep the other one of the pair (include/hnr.h) -- the
    reference's own choice there depends on the order its atomics filled the cell lists in.  What must hold, and is asserted: the same
    distances (the sorted d^2 lists are bit-identical to the oracle's), the same points wherever the distance is below the list's largest
    one, and a point of the right distance in the tied places."""
    from oracle import query_oracle as qo
    from hybridneuralrendering_amd import querier as Q
    rng = np.random.default_rng(3)
    half, _ = scenes.room_cloud(20000, 3, size=(1.0, 0.8, 0.25), n_clutter=3, thickness=0.003)
    half = half[np.abs(half[:, 2]) > 1e-4]
    half[:, 2] = np.abs(half[:, 2])
    xyz = np.ascontiguousarray(np.concatenate([half, half * np.array([1, 1, -1], np.float32)], axis=0).astype(np.float32))
    n_half = half.shape[0]
    hp = qo.hyperparameters(xyz, [0.008] * 3, [2, 2, 2], [3, 3, 3], [-10.0] * 3 + [10.0] * 3, 4.0)
    campos = np.array([-0.3, -0.25, 0.0], np.float32)
    ang = rng.uniform(0.15, 1.2, size=96).astype(np.float32)
    rays = np.ascontiguousarray(np.stack([np.cos(ang), np.sin(ang), np.zeros_like(ang)], axis=1).astype(np.float32))
    cs = dict(xyz=xyz, hp=hp, campos=campos, camrot=np.eye(3, dtype=np.float32), rays=rays, tmid=qo.tmid_table(0.05, 1.5, 200), P=26, max_o=100000, K=8, SR=24)
    og, ref, g, res = _run_both(cs)
    _assert_query_equal(ref, res, cs)                                                      # reference order: slot-exact, ties or not
    d = _dev()
    srt = Q.march_query(g, torch.from_numpy(campos).to(d), torch.from_numpy(rays).to(d), torch.from_numpy(np.ascontiguousarray(cs["tmid"])).to(d),
                        cs["SR"], 8, hp["radius2"], [3, 3, 3], knn_order=1)
    for k in ("ray_nsamp", "sample_loc_w", "ray_mask", "counts"):
        assert torch.equal(srt[k], res[k]), k
    a, b = srt["sample_pidx"].cpu().numpy().reshape(-1, 8), ref["full_pidx"].reshape(-1, 8)
    loc = ref["full_loc"].reshape(-1, 3)
    assert (loc[:, 2] == 0).all()
    def dist2(ids):
        dv = xyz[np.maximum(ids, 0)] - loc[:, None, :]
        d2 = (dv[..., 0] * dv[..., 0] + dv[..., 1] * dv[..., 1]) + dv[..., 2] * dv[..., 2]
        return np.where(ids >= 0, d2, np.float32(np.inf)).astype(np.float32)
    da, db = dist2(a), dist2(b)
    assert ((a >= 0).sum(axis=1) == (b >= 0).sum(axis=1)).all()
    assert (da[:, :-1] <= da[:, 1:]).all()                                                 # ascending
    np.testing.assert_array_equal(da, np.sort(db, axis=1))                                 # the same distances, bit for bit
    kth = np.sort(db, axis=1)[:, -1:]                                                      # the list's largest distance (inf while it is not full)
    inner_a = np.where(da < kth, a, -2); inner_b = np.where(db < kth, b, -2)
    np.testing.assert_array_equal(np.sort(inner_a, axis=1), np.sort(inner_b, axis=1))      # the same points below it
    full = (b >= 0).all(axis=1)
    tied_inside = ((da[:, :-1] == da[:, 1:]) & np.isfinite(da[:, 1:])).any(axis=1)
    partner = np.where(a >= n_half, a - n_half, a + n_half)
    last_alone = full & ~(partner[:, -1:] == a).any(axis=1)                               # a full list whose last point's mirror image is not in it: a tie decided the last place (if the mirror image was a candidate)
    differ = (np.sort(a, axis=1) != np.sort(b, axis=1)).any(axis=1)
    assert tied_inside.sum() > 200 and last_alone.sum() > 20, (int(tied_inside.sum()), int(last_alone.sum()))
    with capsys.disabled():
        print("\n[k-NN ties] %d samples (%d full lists): %d with equidistant neighbours, %d full lists end on one point of a mirror pair, %d sets differ from the "
              "reference-order replay (in tied last places only)" % (a.shape[0], int(full.sum()), int(tied_inside.sum()), int(last_alone.sum()), int(differ.sum())))


def test_sorted_neighbour_order_bad_arguments():
    from hybridneuralrendering_amd import querier as Q
    from hybridneuralrendering_amd._lib import HnrError
    cs = _case(0, 4000, 4, 100000, 4, 6, 64)
    og, ref, g, res = _run_both(cs)
    d = _dev()
    with pytest.raises(HnrError, match="knn_order"):
        Q.march_query(g, torch.from_numpy(cs["campos"]).to(d), torch.from_numpy(cs["rays"]).to(d), torch.from_numpy(np.ascontiguousarray(cs["tmid"])).to(d),
                      cs["SR"], 4, cs["hp"]["radius2"], [3, 3, 3], knn_order=1)


_VARIANT = r'''
import sys, hashlib, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from hybridneuralrendering_amd import scenes, querier as Q
from oracle import query_oracle as qo
xyz, _ = scenes.room_cloud(60000, 11, size=(1.0, 0.8, 0.5), n_clutter=3, thickness=0.003)
hp = qo.hyperparameters(xyz, [0.008] * 3, [2, 2, 2], [3, 3, 3], [-10.0] * 3 + [10.0] * 3, 4.0)
s = np.asarray((1.0, 0.8, 0.5))
cam = scenes.look_at(-0.3 * s + [0, 0, 0.25 * s[2]], 0.4 * s * [1, 1, -0.5])
rays = scenes.camera_rays(scenes.pixel_grid(64, 48), scenes.pinhole(64, 48, 57.6), cam)
d = torch.device("cuda:0")
g = Q.VoxelGrid(torch.from_numpy(xyz).to(d), hp["origin"], hp["cell"], hp["dims"], [3, 3, 3], 12, 500000)
tm = torch.from_numpy(qo.tmid_table(0.05, 1.5, 200)).to(d)
out = []
for order in (0, 1):
    r = Q.march_query(g, torch.from_numpy(cam[:3, 3].copy()).to(d), torch.from_numpy(rays).to(d), tm, 24, 8, hp["radius2"], [3, 3, 3], knn_order=order)
    h = hashlib.sha1()
    for k in ("sample_pidx", "sample_loc_w", "ray_nsamp", "ray_mask", "counts"):
        h.update(r[k].cpu().numpy().tobytes())
    from hybridneuralrendering_amd._lib import CNT
    h.update(r["work"][:int(r["counts"][CNT["SAMPLES"]])].cpu().numpy().tobytes())       # the work list: the kept samples' item ids in ray order
    out.append(h.hexdigest())
print("VARIANT", out[0], out[1], int(g.stats["bytes"]))
'''


def test_every_knn_kernel_variant_returns_the_same_bits(tmp_path):
    """The k-NN has several forms (csrc/query.hip): over the grid's 3x3x3 neighbourhood lists with a quad of lanes per sample (HNR_KNN=4, the default for the
    set-exact order) or one lane per sample (7 / 5: with / without the in-workgroup sort by list length; 8: sorted by cell as well; 9: candidates fetched quad-cooperatively
    through LDS; 6: the quad form in work-list order), and the
    27-cell walk without the lists (HNR_KNN=3, and what runs when the grid carries no lists: HNR_NB_LISTS=0); the march probes a ray's blocks together or one after the other (HNR_MARCH_PROBE=2), or one depth in four with
    the groups near the mask expanded (HNR_MARCH_TWO_LEVEL=1).  Every combination must return the bits the default does -- which the tests above pin to the oracle --
    in both neighbour orders, counters included."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "v.py"
    script.write_text(_VARIANT)

    def run(**env):
        p = subprocess.run([sys.executable, str(script), root], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr[-2000:]
        return [l for l in p.stdout.splitlines() if l.startswith("VARIANT")][0].split()[1:]
    base = run()
    for env in (dict(HNR_KNN="3"), dict(HNR_KNN="4"), dict(HNR_KNN="5"), dict(HNR_KNN="6"), dict(HNR_KNN="7"), dict(HNR_KNN="9"), dict(HNR_KNN="10"), dict(HNR_NB_LISTS="0"), dict(HNR_MARCH_PROBE="2"), dict(HNR_MARCH_TWO_LEVEL="1"), dict(HNR_MARCH_TWO_LEVEL="1", HNR_MARCH_RAYS_PER_WAVE="1"),
                dict(HNR_MARCH_RAYS_PER_WAVE="1"), dict(HNR_MARCH_RAYS_PER_WAVE="64")):
        got = run(**env)
        assert got[:2] == base[:2], (env, got, base)
        if "HNR_NB_LISTS" in env:
            assert int(got[2]) < int(base[2])                      # the lists are what the extra bytes of the handle are
