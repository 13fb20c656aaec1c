"""Synthetic stand-ins for the BASELINE.json configs.

No dataset or checkpoint of the reference is available offline, so every config is
restated as a seeded synthetic scene carrying the scene's hyper-parameters from the
reference launch scripts (dev_scripts/w_scannet_etf/scene241_hybrid.sh:48-61,
dev_scripts/w_n360/lego_hybrid.sh, chair_hybrid.sh, scene101_full.sh).

Pure numpy; used by tests/, bench.py and __graft_entry__.smoke().
"""
from types import SimpleNamespace

import numpy as np


# ----------------------------------------------------------------------------- options
def default_opt(**kw):
    """Hot-path flags at the values shared by all 19 shipped launch scripts."""
    opt = SimpleNamespace(
        # query (models/neural_points/neural_points.py:13-230)
        wcoord_query=1, NN=2, K=8, SR=24, P=26, max_o=610000,
        vsize=[0.008, 0.008, 0.008], vscale=[2, 2, 2], kernel_size=[3, 3, 3], query_size=[3, 3, 3],
        radius_limit_scale=4.0, depth_limit_scale=0.0, ranges=[-10.0, -10.0, -10.0, 10.0, 10.0, 10.0],
        z_depth_dim=400, inverse=0, gpu_maxthr=1024, is_train=0, load_points=1,
        point_features_dim=32, point_conf_mode="1", point_dir_mode="1", point_color_mode="1",
        xyz_grad=0, feat_grad=1, conf_grad=1, dir_grad=1, color_grad=1, default_conf=-1.0,
        construct_res=0, grid_res=0, point_noise="", num_point=8192, cloud_path="", feedforward=0,
        # aggregator (models/aggregators/point_aggregators.py:28-425)
        which_agg_model="viewmlp", agg_distance_kernel="linear", agg_intrp_order=2, agg_dist_pers=20,
        agg_weight_norm=1, agg_axis_weight=None, apply_pnt_mask=1,
        agg_feat_xyz_mode="None", agg_alpha_xyz_mode="None", agg_color_xyz_mode="None",
        act_type="LeakyReLU", act_super=1, shading_feature_num=256,
        shading_feature_mlp_layer0=1, shading_feature_mlp_layer1=2, shading_feature_mlp_layer2=0,
        shading_feature_mlp_layer3=2, shading_alpha_mlp_layer=1, shading_color_mlp_layer=4,
        shading_color_channel_num=3, num_feat_freqs=3, dist_xyz_freq=5, dist_xyz_deno=0.0,
        num_pos_freqs=10, num_viewdir_freqs=4, view_ori=0, point_hyper_dim=256,
        weight_xyz_freq=2, weight_feat_dim=8, sh_degree=4,
        use_nearest=4, dynamic_nearest=0, use_delta_view=1, feature_guidance=1, mixup_mode="partial",
        learn_residuals=1, dynamic_weight=0, refine_blend=0, tradition_attention=0, add_idx=0,
        drop_ratio=0.5, drop_patch=1, ray_points=1, random_position=1, drop_disturb_range=0,
        dilation_setup="7_8_1_8", downweight_blurry_feats=0, disable_viewdirs=0, disable_color_feature=0,
        separate_color_decoder=0, large_color_final_block=0, use_2D_CNN=0, learnable_blur_kernel=0,
        learnable_blur_kernel_conv=0, learnable_blur_kernel_size=9, learnable_blur_patch_size=8,
        learnable_blur_kernel_mode=4, learnable_blur_kernel_norm=0, search_size=0, search_dilation=0, exp_aggregation=0,
        sparse_loss_weight=0, zero_one_loss_items=["conf_coefficient"], prob=0,
        # render shell (models/neural_points_volumetric_model.py:47-70)
        raydist_mode_unit=1, which_render_func="radiance", which_blend_func="alpha",
        which_tonemap_func="off", near_plane=0.1, far_plane=8.0,
    )
    for k, v in kw.items():
        setattr(opt, k, v)
    return opt


SCENE_OPTS = {
    # chair_hybrid.sh
    "chair": dict(vsize=[0.004] * 3, SR=80, P=12, max_o=410000,
                  ranges=[-0.721, -0.695, -0.995, 0.658, 0.706, 1.050], near_plane=2.0, far_plane=6.0),
    # lego_hybrid.sh
    "lego": dict(vsize=[0.004] * 3, SR=80, P=9, max_o=830000,
                 ranges=[-0.638, -1.141, -0.346, 0.634, 1.149, 1.141], near_plane=2.0, far_plane=6.0),
    # scene241_hybrid.sh
    "scene0241": dict(vsize=[0.008] * 3, SR=24, P=26, max_o=610000, ranges=[-10.0] * 3 + [10.0] * 3,
                      near_plane=0.1, far_plane=8.0),
    # scene101_full.sh
    "scene0101": dict(vsize=[0.008] * 3, SR=24, P=30, max_o=2000000, ranges=[-10.0] * 3 + [10.0] * 3,
                      near_plane=0.1, far_plane=8.0),
}


ROOM_SIZE = {"scene0241": (5.4, 4.1, 2.5), "scene0101": (8.0, 6.0, 3.0)}


def scene_opt(name, **kw):
    d = dict(SCENE_OPTS[name])
    d.update(kw)
    return default_opt(**d)


# ----------------------------------------------------------------------------- clouds
def _box_surface(rng, n, lo, hi, inward=True):
    """n points uniform on the 6 faces of the axis-aligned box [lo,hi]; returns (xyz, normal)."""
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    ext = hi - lo
    areas = np.array([ext[1] * ext[2], ext[1] * ext[2], ext[0] * ext[2], ext[0] * ext[2],
                      ext[0] * ext[1], ext[0] * ext[1]])
    face = rng.choice(6, size=n, p=areas / areas.sum())
    p = lo + rng.random((n, 3)) * ext
    axis = face // 2
    side = face % 2
    p[np.arange(n), axis] = np.where(side == 0, lo[axis], hi[axis])
    nrm = np.zeros((n, 3))
    sign = np.where(side == 0, 1.0, -1.0) * (1.0 if inward else -1.0)
    nrm[np.arange(n), axis] = sign
    return p, nrm


def room_cloud(n, seed, size=(8.0, 6.0, 3.0), clutter_frac=0.25, n_clutter=12, thickness=0.004):
    """ScanNet-like cloud: inner faces of a size[0] x size[1] x size[2] m room centred at the
    origin (floor at z=-size[2]/2) plus box clutter standing on the floor.  Returns xyz, normals."""
    rng = np.random.default_rng(seed)
    half = np.asarray(size) / 2
    n_cl = int(n * clutter_frac)
    n_room = n - n_cl
    xyz, nrm = _box_surface(rng, n_room, -half, half, inward=True)
    parts_xyz, parts_nrm = [xyz], [nrm]
    if n_cl > 0:
        sc = min(size[0], size[1]) / 6.0                     # clutter scales with the room
        sizes = rng.uniform(0.3, 1.4, size=(n_clutter, 3)) * sc
        sizes[:, 2] = rng.uniform(0.1, 0.55, size=n_clutter) * size[2]
        ctr = np.stack([rng.uniform(-half[0] + 0.8 * sc, half[0] - 0.8 * sc, n_clutter),
                        rng.uniform(-half[1] + 0.8 * sc, half[1] - 0.8 * sc, n_clutter),
                        -half[2] + sizes[:, 2] / 2], axis=1)
        area = 2 * (sizes[:, 0] * sizes[:, 1] + sizes[:, 1] * sizes[:, 2] + sizes[:, 0] * sizes[:, 2])
        cnt = np.floor(n_cl * area / area.sum()).astype(int)
        cnt[0] += n_cl - cnt.sum()
        for i in range(n_clutter):
            p, q = _box_surface(rng, int(cnt[i]), ctr[i] - sizes[i] / 2, ctr[i] + sizes[i] / 2, inward=False)
            parts_xyz.append(p); parts_nrm.append(q)
    xyz = np.concatenate(parts_xyz); nrm = np.concatenate(parts_nrm)
    xyz = xyz + nrm * rng.normal(0.0, thickness, size=(xyz.shape[0], 1))
    perm = rng.permutation(xyz.shape[0])
    return xyz[perm].astype(np.float32), nrm[perm].astype(np.float32)


def object_cloud(n, seed, ranges, n_boxes=9, thickness=0.002):
    """NeRF-synthetic-like cloud: surfaces of a union of boxes filling `ranges` (6 floats)."""
    rng = np.random.default_rng(seed)
    lo, hi = np.asarray(ranges[:3], np.float64), np.asarray(ranges[3:], np.float64)
    ext = hi - lo
    sizes = rng.uniform(0.15, 0.6, size=(n_boxes, 3)) * ext
    ctr = lo + sizes / 2 + rng.random((n_boxes, 3)) * (ext - sizes)
    area = 2 * (sizes[:, 0] * sizes[:, 1] + sizes[:, 1] * sizes[:, 2] + sizes[:, 0] * sizes[:, 2])
    cnt = np.floor(n * area / area.sum()).astype(int)
    cnt[0] += n - cnt.sum()
    px, pn = [], []
    for i in range(n_boxes):
        p, q = _box_surface(rng, int(cnt[i]), ctr[i] - sizes[i] / 2, ctr[i] + sizes[i] / 2, inward=False)
        px.append(p); pn.append(q)
    xyz = np.concatenate(px); nrm = np.concatenate(pn)
    xyz = xyz + nrm * rng.normal(0.0, thickness, size=(xyz.shape[0], 1))
    xyz = np.clip(xyz, lo + 1e-4, hi - 1e-4)
    perm = rng.permutation(xyz.shape[0])
    return xyz[perm].astype(np.float32), nrm[perm].astype(np.float32)


def point_attributes(normals, seed, feat_dim=32):
    """embedding N(0,0.3^2) [1,N,F], conf U[0.1,1] [1,N,1], dir = normals [1,N,3], color U[0,1] [1,N,3]."""
    rng = np.random.default_rng(seed + 1000)
    n = normals.shape[0]
    emb = rng.normal(0.0, 0.3, size=(1, n, feat_dim)).astype(np.float32)
    conf = rng.uniform(0.1, 1.0, size=(1, n, 1)).astype(np.float32)
    color = rng.uniform(0.0, 1.0, size=(1, n, 3)).astype(np.float32)
    return emb, conf, normals[None].astype(np.float32), color


# ----------------------------------------------------------------------------- cameras
def look_at(eye, target, up=(0.0, 0.0, 1.0)):
    """c2w [4,4] in the reference's camera convention (x right, y down, z forward;
    data/data_utils.py:58-72 builds rays as [x, y, 1] @ rot.T)."""
    eye, target, up = (np.asarray(v, np.float64) for v in (eye, target, up))
    z = target - eye
    z /= np.linalg.norm(z)
    x = np.cross(z, up)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = x, y, z, eye
    return c2w.astype(np.float32)


def pinhole(w, h, focal):
    return np.array([[focal, 0, w / 2], [0, focal, h / 2], [0, 0, 1]], dtype=np.float32)


def pixel_grid(w, h, margin=0):
    """All (px, py) of the frame minus a `margin` border, scan-line order
    (data/scannet_ft_dataset.py:946-949 'no_crop')."""
    px, py = np.meshgrid(np.arange(margin, w - margin).astype(np.int32),
                         np.arange(margin, h - margin).astype(np.int32))
    return np.stack([px, py], axis=-1).reshape(-1, 2)


def camera_rays(pixels, intrinsic, c2w):
    """get_dtu_raydir (data/data_utils.py:58-72) with dir_norm=0: raydir [R,3] float32."""
    x = (pixels[..., 0] + 0.5 - intrinsic[0, 2]) / intrinsic[0, 0]
    y = (pixels[..., 1] + 0.5 - intrinsic[1, 2]) / intrinsic[1, 1]
    z = np.ones_like(x)
    dirs = np.stack([x, y, z], axis=-1)
    dirs = dirs @ c2w[:3, :3].T
    return np.reshape(dirs, (-1, 3)).astype(np.float32)


def reference_images(n_views, h, w, seed):
    """Smooth random RGB images in [0,1], [n_views, h, w, 3] float32 (stand-in for images_nearest)."""
    rng = np.random.default_rng(seed + 2000)
    yy, xx = np.meshgrid(np.linspace(0, 1, h), np.linspace(0, 1, w), indexing="ij")
    imgs = np.zeros((n_views, h, w, 3), np.float32)
    for v in range(n_views):
        for c in range(3):
            acc = np.zeros((h, w))
            for _ in range(6):
                fx, fy = rng.uniform(1, 12, size=2)
                ph = rng.uniform(0, 2 * np.pi, size=2)
                acc += rng.uniform(0.3, 1.0) * np.sin(2 * np.pi * fx * xx + ph[0]) * np.cos(2 * np.pi * fy * yy + ph[1])
            acc = (acc - acc.min()) / (acc.max() - acc.min() + 1e-9)
            imgs[v, :, :, c] = np.round(acc * 255) / 255          # 8-bit images, like the decoded JPEG frames
    return imgs


def make_scene(name, n_points, seed, w=None, h=None, n_views=4, size=None):
    """Bundle: opt, cloud + attributes, a camera inside/around the scene, 4 neighbouring reference
    cameras and images.  name in SCENE_OPTS."""
    opt = scene_opt(name)
    if name in ("chair", "lego"):
        xyz, nrm = object_cloud(n_points, seed, opt.ranges)
        w = w or (200 if name == "chair" else 800)
        h = h or w
        focal = 0.5 * w / np.tan(0.5 * 0.6911112070083618)     # nerf_synth360_ft_dataset.py:155-162
        ctr = (np.asarray(opt.ranges[:3]) + np.asarray(opt.ranges[3:])) / 2
        eye = ctr + np.array([2.6, 2.4, 1.6]) * (4.0 / np.linalg.norm([2.6, 2.4, 1.6]))
        cams = [look_at(eye, ctr)]
        for k in range(n_views):
            ang = 0.08 * (k - (n_views - 1) / 2 + 0.5)
            rot = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
            cams.append(look_at(ctr + rot @ (eye - ctr), ctr))
    else:
        # Room sized so that the occupied-voxel count of a 2 M (4 M) point cloud stays below the scene's
        # max_o, as it does for the real scans (scene0241: ~0.55 M of 0.61 M; scene0101: ~1.0 M of 2.0 M).
        size = size or ROOM_SIZE[name]
        xyz, nrm = room_cloud(n_points, seed, size=size, clutter_frac=0.2, thickness=0.003)
        w, h = w or 640, h or 480
        focal = 577.87 * w / 640.0                                # ScanNet colour intrinsics, scaled
        sz = np.asarray(size)
        eye = sz * np.array([-0.33, -0.28, 0.03])
        tgt = sz * np.array([0.31, 0.27, -0.2])
        cams = [look_at(eye, tgt)]
        for k in range(n_views):
            off = np.array([0.06, -0.05, 0.01]) * (k - (n_views - 1) / 2 + 0.5) * 2
            cams.append(look_at(eye + off, tgt + 0.5 * off))
    emb, conf, pdir, color = point_attributes(nrm, seed, feat_dim=opt.point_features_dim)
    K = pinhole(w, h, focal)
    return SimpleNamespace(
        name=name, opt=opt, xyz=xyz, normals=nrm, emb=emb, conf=conf, dir=pdir, color=color,
        w=w, h=h, intrinsic=K, c2w=cams[0], c2w_nearest=np.stack(cams[1:1 + n_views]),
        images_nearest=reference_images(n_views, h, w, seed),
        near=float(opt.near_plane), far=float(opt.far_plane), bg_color=np.ones(3, np.float32))


# ----------------------------------------------------------------------------- training batches (configs C3 / C5)
def dilated_patch_batch(w, h, margin, dilation_setup, seed):
    """Pixel coordinates [S*S, 2] (x, y) of one `random_sample='dilated'` batch (data/scannet_ft_dataset.py:918-949 in the reference):
    patch_num x patch_num patches of patch_size x patch_size pixels, each with its own stride in [d_lo, d_hi] and position, laid out as
    an S x S grid (S = patch_num * patch_size) in row-major ray order.  dilation_setup = "7_8_1_6"."""
    pn, ps, d_lo, d_hi = (int(v) for v in dilation_setup.split("_"))
    rng = np.random.default_rng(seed)
    S = pn * ps
    px, py = np.zeros((S, S), np.int32), np.zeros((S, S), np.int32)
    gx, gy = np.meshgrid(np.arange(ps), np.arange(ps))
    for pi in range(pn):
        for pj in range(pn):
            d = int(rng.integers(d_lo, d_hi + 1))
            x0 = int(rng.integers(margin, w - margin - (ps - 1) * d))
            y0 = int(rng.integers(margin, h - margin - (ps - 1) * d))
            px[pi * ps:(pi + 1) * ps, pj * ps:(pj + 1) * ps] = x0 + d * gx
            py[pi * ps:(pi + 1) * ps, pj * ps:(pj + 1) * ps] = y0 + d * gy
    return np.stack([px, py], axis=-1).reshape(-1, 2).astype(np.int32), pn, ps


def blur_kernels_v2(k_size=9, dists=(1, 2, 4), n_dirs=8):
    """`blur_kernel_version=2` of the reference (data/scannet_ft_dataset.py:214-242): for every motion length in `dists` and
    every one of the n_dirs / 2 directions (0, 45, 90, 135 degrees) a normalised symmetric line kernel -> [12, 9, 9]."""
    c = k_size // 2
    dirs = [(0, 1), (1, 1), (1, 0), (1, -1)][:n_dirs // 2]
    out = []
    for dist in dists:
        for dy, dx in dirs:
            k = np.zeros((k_size, k_size), np.float32)
            for t in range(-dist, dist + 1):
                k[c + t * dy, c + t * dx] = 1.0
            out.append(k / k.sum())
    return np.stack(out)
