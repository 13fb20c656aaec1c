"""CPU restatement (torch fp32) of the gather / aggregate / composite half of the hot path.
TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this.

PINNED: tests/test_render_oracle.py checks every function here against tests/golden/render_*.npz,
which tests/golden/make_golden.py produced by running the imported reference
(NeuralPointsRayMarching.forward + fill_invalid) on CPU.

Restated for the one configuration all 19 shipped launch scripts use:
which_agg_model=viewmlp, agg_intrp_order=2, agg_distance_kernel=linear, agg_dist_pers=20,
agg_weight_norm=1, apply_pnt_mask=1, num_feat_freqs=3, dist_xyz_freq=5, dist_xyz_deno=0,
num_viewdir_freqs=4, view_ori=0, point_{conf,dir,color}_mode="1", act_type=LeakyReLU, act_super=1,
use_nearest=4, feature_guidance=1, use_delta_view=1, mixup_mode=partial, learn_residuals=1,
raydist_mode_unit=1, radiance render / alpha blend / no tone map.  Rw2c = identity.

Reference files (relative to /root/reference):
  models/neural_points/neural_points.py            :607-613 (w2pers), :709-733 (gather)
  models/neural_points_volumetric_model.py         :248-255 (w2iproject), :296-310 (delta dirs),
                                                   :331-339 (ray_dist), :353-391 (outputs), :87-126 (fill_invalid)
  models/aggregators/point_aggregators.py          :1427-1522 (forward), :825-833 (linear), :892-1338 (viewmlp)
  models/helpers/networks.py                       :175-189 (positional_encoding)
  models/rendering/diff_ray_marching.py            :508-557 (ray_march)
"""
import numpy as np
import torch
import torch.nn.functional as F


def positional_encoding(positions, freqs, ori=False):
    """networks.py:175-189."""
    freq_bands = (2 ** torch.arange(freqs).float())
    ori_c = positions.shape[-1]
    pts = (positions[..., None] * freq_bands).reshape(positions.shape[:-1] + (freqs * positions.shape[-1],))
    if ori:
        return torch.cat([positions, torch.sin(pts), torch.cos(pts)], dim=-1).reshape(pts.shape[:-1] + (pts.shape[-1] * 2 + ori_c,))
    return torch.stack([torch.sin(pts), torch.cos(pts)], dim=-1).reshape(pts.shape[:-1] + (pts.shape[-1] * 2,))


def w2pers_points(xyz, camrotc2w, campos):
    """neural_points.py:607-613.  xyz [N,3], camrotc2w [1,3,3], campos [1,3] -> [1,N,3]."""
    shift = xyz[None, ...] - campos[:, None, :]
    c = torch.sum(camrotc2w[:, None, :, :] * shift[:, :, :, None], dim=-2)
    return torch.stack([c[:, :, 0] / c[:, :, 2], c[:, :, 1] / c[:, :, 2], c[:, :, 2]], dim=-1)


def w2pers_samples(loc_w, camrotc2w, campos):
    """query_point_indices_worldcoords.py:96-103.  loc_w [1,R,SR,3]."""
    shift = loc_w - campos[:, None, :]
    c = torch.sum(shift[..., None, :] * torch.transpose(camrotc2w, 1, 2)[:, None, None, ...], dim=-1)
    return torch.stack([c[..., 0] / c[..., 2], c[..., 1] / c[..., 2], c[..., 2]], dim=-1)


def gather_points(xyz, emb, conf, pdir, color, sample_pidx, camrotc2w, campos):
    """neural_points.py:709-720.  sample_pidx [1,R,SR,K] int (-1 = empty).  Empty slots read point 0
    (clamp(min=0)); they are masked later by sample_pnt_mask."""
    pers = w2pers_points(xyz, camrotc2w, campos)
    mask = sample_pidx >= 0
    B, R, SR, K = sample_pidx.shape
    idx = torch.clamp(sample_pidx, min=0).view(-1).long()
    cat = torch.cat([xyz[None, ...], pers, emb], dim=-1)
    se = torch.index_select(cat, 1, idx).view(B, R, SR, K, emb.shape[2] + 6)
    sc = torch.index_select(color, 1, idx).view(B, R, SR, K, 3)
    sd = torch.index_select(pdir, 1, idx).view(B, R, SR, K, 3)
    scf = torch.index_select(conf, 1, idx).view(B, R, SR, K, 1)
    return dict(sampled_color=sc, sampled_dir=sd, sampled_conf=scf, sampled_embedding=se[..., 6:],
                sampled_xyz_pers=se[..., 3:6], sampled_xyz=se[..., :3], sample_pnt_mask=mask)


def w2iproject(sample_loc_w, intrinsic, c2w):
    """neural_points_volumetric_model.py:248-255.  sample_loc_w [R,SR,3] -> [R,SR,2]."""
    h = torch.cat([sample_loc_w, torch.ones_like(sample_loc_w[..., :1])], dim=-1)
    w2c = torch.inverse(c2w).t()
    c = h @ w2c
    i = c[:, :, :3] @ intrinsic.t()
    d = i[:, :, 2:3]
    return (i / (d + 1e-10))[:, :, 0:2]


def project_nearest(sample_loc_w, campos, c2w_nearest, campos_nearest, intrinsic_nearest):
    """:287-310.  sample_loc_w [1,R,SR,3] -> sample_loc_i_n [V,R,SR,2], delta_viewdir_n [V,R,SR,3]."""
    V = c2w_nearest.shape[1]
    loc_i = torch.stack([w2iproject(sample_loc_w[0], intrinsic_nearest[0], c2w_nearest[0, v]) for v in range(V)])
    cur = sample_loc_w - campos[0]
    cur = cur / (torch.linalg.norm(cur, dim=-1, keepdim=True) + 1e-6)
    dl = []
    for v in range(V):
        nv = sample_loc_w - campos_nearest[0, v, :]
        nv = nv / (torch.linalg.norm(nv, dim=-1, keepdim=True) + 1e-6)
        dl.append((nv - cur)[0])
    return loc_i, torch.stack(dl)


def gathered_pixels(sample_loc_w, c2w_nearest, intrinsic_nearest, H, W):
    """The integer pixel (px, py) of every reference view that `aggregate` gathers for world positions sample_loc_w [R', SR, 3] (w2iproject :248-255, then
    `.to(torch.int32)` + the bounds rule, point_aggregators.py:1077-1088); (-1, -1) where the rule masks the row.  Returns int64 [V, ..., 2].
    Test infrastructure for the whole-frame check of bench.py: which rays gather the same pixels in the oracle and in the HIP path."""
    V = c2w_nearest.shape[1]
    out = []
    for v in range(V):
        li = w2iproject(sample_loc_w, intrinsic_nearest[0], c2w_nearest[0, v]).reshape(-1, 2)      # the shape `render` projects ([R', SR, 3]): same BLAS path
        px, py = li[:, 0].to(torch.int32), li[:, 1].to(torch.int32)
        inval = (px < 0) | (px >= W) | (py < 0) | (py >= H)
        out.append(torch.stack([torch.where(inval, torch.full_like(px, -1), px), torch.where(inval, torch.full_like(py, -1), py)], dim=-1).long())
    return torch.stack(out).reshape((V,) + tuple(sample_loc_w.shape[:-1]) + (2,))


def _seq(x, sd, name, idxs, act_last=True, slope=0.01):
    """nn.Sequential of Linear(+LeakyReLU) with the reference's parameter names `<name>.<i>.weight`."""
    for j, i in enumerate(idxs):
        x = F.linear(x, sd["%s.%d.weight" % (name, i)], sd["%s.%d.bias" % (name, i)])
        if act_last or j < len(idxs) - 1:
            x = F.leaky_relu(x, slope)
    return x


def image_features(images_nearest, sd, slope=0.01):
    """point_aggregators.py:1047-1067, :1089.  images_nearest [1,V,H,W,3] -> [V,45,H,W] with pixel (0,0) zeroed."""
    img = images_nearest[0].permute(0, 3, 1, 2)

    def blk(x, name, stride):
        x = F.leaky_relu(F.conv2d(x, sd[name + ".0.weight"], sd[name + ".0.bias"], stride=stride, padding=1), slope)
        return F.leaky_relu(F.conv2d(x, sd[name + ".2.weight"], sd[name + ".2.bias"], stride=1, padding=1), slope)

    s1 = blk(img, "aux_block_s1", 2)
    s2 = blk(s1, "aux_block_s2", 2)
    s3 = blk(s2, "aux_block_s3", 2)
    H, W = img.shape[2], img.shape[3]
    out = torch.cat([img, F.interpolate(s1, size=[H, W], mode="bilinear"), F.interpolate(s2, size=[H, W], mode="bilinear"),
                     F.interpolate(s3, size=[H, W], mode="bilinear")], dim=1).clone()
    out[:, :, 0, 0] = out[:, :, 0, 0] * 0.0
    return out


def aggregate(g, sample_loc, sample_loc_w, sample_ray_dirs, sd, loc_i_n, delta_viewdir_n, images_nearest,
              is_train=False, drop_ray_rows=None, frame_weight_n=None, use_nearest=4):
    """PointAggregator.forward (:1427-1522) + viewmlp (:892-1338), order-2 hybrid path.

    g = gather_points(...) dict.  Returns dict(decoded [1,R,SR,4], ray_valid [1,R,SR] bool, weight [1,R,SR,K],
    conf_coefficient [1,R,SR,K]).  drop_ray_rows: rows (valid-ray index) whose image feature is zeroed at
    train time (drop_patch_rays pattern, :1222-1237)."""
    mask = g["sample_pnt_mask"]
    B, R, SR, K = mask.shape
    ray_valid = torch.any(mask, dim=-1).view(-1)
    total = ray_valid.numel()
    if total == 0 or int(ray_valid.sum()) == 0:
        return dict(decoded=torch.zeros((B, R, SR, 4)), ray_valid=ray_valid.view(B, R, SR), weight=None, conf_coefficient=None)
    xp, xw = g["sampled_xyz_pers"], g["sampled_xyz"]
    # agg_dist_pers == 20 (:1472-1480)
    xd = xp[..., 0] * xp[..., 2] - sample_loc[:, :, :, None, 0] * sample_loc[:, :, :, None, 2]
    yd = xp[..., 1] * xp[..., 2] - sample_loc[:, :, :, None, 1] * sample_loc[:, :, :, None, 2]
    zd = xp[..., 2] - sample_loc[:, :, :, None, 2]
    dists = torch.cat([xw - sample_loc_w[..., None, :], torch.stack([xd, yd, zd], dim=-1)], dim=-1)
    # linear kernel (:825-833) + normalisation (:1500-1501)
    weight = mask * (1. / torch.clamp(torch.norm(dists[..., :3], dim=-1), min=1e-6))
    weight = weight / torch.clamp(torch.sum(weight, dim=-1, keepdim=True), min=1e-8)
    sc0 = g["sampled_conf"][..., 0]                                              # gradiant_clamp (:1422-1424): clamp forward,
    conf_c = sc0 - (sc0 - torch.clamp(sc0, min=0.0001, max=1)).detach()          # identity backward
    w_agg = (weight * conf_c).view(B * R * SR, K, 1)

    pm = mask.view(-1)
    viewdirs = sample_ray_dirs.view(-1, 3)                                       # Rw2c = I (:908)
    pe_v = positional_encoding(viewdirs, 4, ori=True)
    ori_viewdirs, vd = pe_v[..., :3], pe_v[..., 3:]
    vd = vd[ray_valid, :]
    # per-neighbour rows (:922-939)
    dists_flat = dists.view(-1, 6)[pm, :]
    dists_pe = positional_encoding(dists_flat, 5)
    feat = g["sampled_embedding"].reshape(-1, g["sampled_embedding"].shape[-1])[pm, :]
    feat = torch.cat([feat, positional_encoding(feat, 3)], dim=-1)
    feat = torch.cat([feat, dists_pe], dim=-1)
    feat = _seq(feat, sd, "block1", (0, 2))
    # block3 input (:957-972)
    col = g["sampled_color"].reshape(-1, 3)[pm, :]
    pdir = g["sampled_dir"].reshape(-1, 3)[pm, :]
    ov = ori_viewdirs[..., None, :].repeat(1, K, 1).view(-1, 3)[pm, :]
    feat = torch.cat([feat, col, pdir - ov, torch.sum(pdir * ov, dim=-1, keepdim=True)], dim=-1)
    feat = _seq(feat, sd, "block3", (0, 2))
    # alpha (:1002-1014): softplus(x - 1), K-weighted sum
    alpha = F.softplus(F.linear(feat, sd["alpha_branch.0.weight"], sd["alpha_branch.0.bias"]) - 1)
    ah = torch.zeros([B * R * SR * K, 1])
    ah[pm, :] = alpha
    alpha = torch.sum(ah.view(B * R * SR, K, 1) * w_agg, dim=-2).view(-1, 1)[ray_valid, :]
    fh = torch.zeros([B * R * SR * K, feat.shape[-1]])
    fh[pm, :] = feat
    feat = torch.sum(fh.view(B * R * SR, K, -1) * w_agg, dim=-2).view(-1, fh.shape[-1])[ray_valid, :]
    # colour-feature branch (:1028-1037)
    cf = _seq(torch.cat([feat, vd], dim=-1), sd, "color_feature_branch", (0, 2, 4))
    # image branch (:1047-1217); use_nearest == 0 switches it off: merged = 0 (:1257-1258)
    if use_nearest == 0:
        C = 45
        merged = torch.zeros_like(cf)[:, :C]
        ci, cv = cf[:, :C], cf[:, C:]
        mix = _seq(torch.cat((ci, merged), dim=-1), sd, "color_mixup_block", (0, 2, 4), act_last=False) + ci
        rgb = torch.sigmoid(F.linear(torch.cat([mix, cv], dim=-1), sd["color_final_block.0.weight"], sd["color_final_block.0.bias"]))
        rgb = rgb * (1 + 2 * 0.001) - 0.001
        out = torch.zeros([total, 4])
        out[ray_valid] = torch.cat([alpha, rgb], dim=-1)
        return dict(decoded=out.view(B, R, SR, 4), ray_valid=ray_valid.view(B, R, SR), weight=weight, conf_coefficient=conf_c)
    aux = image_features(images_nearest, sd)
    V, C, H1, W1 = aux.shape
    li = loc_i_n.view(V, -1, 2)[:, ray_valid, :].reshape(-1, 2)
    px = li[:, 0].to(torch.int32)                                                # truncation toward zero (:1077-1078)
    py = li[:, 1].to(torch.int32)
    dv = delta_viewdir_n.view(V, -1, 3)[:, ray_valid, :]
    inval = (px < 0) | (px >= W1) | (py < 0) | (py >= H1)
    px = torch.where(inval, torch.zeros_like(px), px).view(V, -1).long()
    py = torch.where(inval, torch.zeros_like(py), py).view(V, -1).long()
    vmask = (~inval).to(torch.float32).view(V, -1)
    wsum, fsum = 0, 0
    for v in range(V):
        f = aux[v:v + 1, :, py[v], px[v]].permute(0, 2, 1).reshape(-1, C)
        wv = torch.sigmoid(_seq(torch.cat((f, cf, dv[v]), dim=-1), sd, "aux_merge_weight_block", (0, 2, 4, 6), act_last=False)) * vmask[v][..., None]
        if frame_weight_n is not None:
            wv = wv * frame_weight_n[0, v]
        fsum = fsum + f * wv
        wsum = wsum + wv
    merged = fsum / (wsum + 1e-6)
    if is_train and drop_ray_rows is not None:                                   # (:1222-1237)
        flag = np.zeros((R, SR))
        flag[[r for r in drop_ray_rows if r < R], :] = 1
        pos = np.where(flag.flatten()[ray_valid.numpy()] == 1)[0]
        merged[pos, :] = merged[pos, :] * 0
    # mix-up (:1285-1295) and final colour (:1334, :478-482)
    ci, cv = cf[:, :C], cf[:, C:]
    mix = _seq(torch.cat((ci, merged), dim=-1), sd, "color_mixup_block", (0, 2, 4), act_last=False) + ci
    rgb = torch.sigmoid(F.linear(torch.cat([mix, cv], dim=-1), sd["color_final_block.0.weight"], sd["color_final_block.0.bias"]))
    rgb = rgb * (1 + 2 * 0.001) - 0.001
    out = torch.zeros([total, 4])
    out[ray_valid] = torch.cat([alpha, rgb], dim=-1)
    return dict(decoded=out.view(B, R, SR, 4), ray_valid=ray_valid.view(B, R, SR), weight=weight, conf_coefficient=conf_c)


def ray_dist(sample_loc, ray_valid, vsize_z, raydist_mode_unit=1):
    """neural_points_volumetric_model.py:331-339."""
    rd = torch.cummax(sample_loc[..., 2], dim=-1)[0]
    rd = torch.cat([rd[..., 1:] - rd[..., :-1], torch.full((rd.shape[0], rd.shape[1], 1), vsize_z)], dim=-1)
    m = rd < 1e-8
    if raydist_mode_unit > 0:
        m = torch.logical_or(m, rd > 2 * vsize_z)
    m = m.to(torch.float32)
    rd = rd * (1.0 - m) + m * vsize_z
    return rd * ray_valid.float()


def ray_march(rd, ray_valid, feats, bg_color):
    """diff_ray_marching.py:508-557 with radiance_render / alpha_blend (diff_render_func.py:36,48)."""
    point_color = feats[..., 1:]
    sigma = feats[..., 0] * ray_valid.float()
    opacity = 1 - torch.exp(-sigma * rd)
    acc = torch.cumprod(1. - opacity + 1e-10, dim=-1)
    bg_t = acc[:, :, [-1]]
    acc = torch.cat([torch.ones(opacity.shape[0:2] + (1,)), acc[:, :, :-1]], dim=-1)
    bw = (opacity * acc)[..., None]
    ray_color = torch.sum(point_color * bw, dim=-2)
    if bg_color is not None:
        ray_color = ray_color + bg_color.float().view(bg_t.shape[0], 1, 3) * bg_t
    return dict(ray_color=ray_color, opacity=opacity, acc_transmission=acc, blend_weight=bw, background_transmission=bg_t)


def render(xyz, emb, conf, pdir, color, sd, q, campos, camrotc2w, raydir_all, bg_color, c2w_nearest, campos_nearest,
           intrinsic_nearest, images_nearest, vsize, raydist_mode_unit=1, is_train=False, drop_ray_rows=None, use_nearest=4, frame_weight_n=None):
    """NeuralPointsRayMarching.forward (:257-391) after the query, + fill_invalid (:87-126).  frame_weight_n [1,V]: downweight_blurry_feats.

    q: dict(sample_pidx [R',SR,K], sample_loc_w [R',SR,3], ray_mask [R]) numpy or tensors (the query 7-tuple core).
    All other arguments are torch CPU tensors shaped like the reference's inputs (leading batch dim 1)."""
    t = lambda a: a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    pidx, loc_w, ray_mask = t(q["sample_pidx"])[None], t(q["sample_loc_w"])[None], t(q["ray_mask"])[None]
    SR = pidx.shape[2]
    sample_loc = w2pers_samples(loc_w, camrotc2w, campos)
    dirs = torch.masked_select(raydir_all, ray_mask[..., None] > 0).reshape(1, -1, 3)[..., None, :].expand(-1, -1, SR, -1).contiguous()
    g = gather_points(xyz, emb, conf, pdir, color, pidx, camrotc2w, campos)
    loc_i, dvd = project_nearest(loc_w, campos, c2w_nearest, campos_nearest, intrinsic_nearest)
    a = aggregate(g, sample_loc, loc_w, dirs, sd, loc_i, dvd, images_nearest, is_train=is_train, drop_ray_rows=drop_ray_rows,
                  use_nearest=use_nearest, frame_weight_n=frame_weight_n)
    rd = ray_dist(sample_loc, a["ray_valid"], vsize[2], raydist_mode_unit)
    m = ray_march(rd, a["ray_valid"], a["decoded"], bg_color)
    out = dict(coarse_raycolor=m["ray_color"], coarse_point_opacity=m["opacity"], coarse_is_background=m["background_transmission"],
               queried_shading=torch.logical_not(torch.any(a["ray_valid"], dim=-1, keepdims=True)).repeat(1, 1, 3).to(torch.float32),
               ray_mask=ray_mask, decoded_features=a["decoded"], ray_valid=a["ray_valid"], weight=a["weight"],
               conf_coefficient=a["conf_coefficient"], blend_weight=m["blend_weight"], sample_loc=sample_loc, ray_dist=rd)
    # fill_invalid (:87-126)
    B, OR = ray_mask.shape
    inds = torch.nonzero(ray_mask)
    isbg = torch.ones([B, OR, 1])
    isbg[inds[..., 0], inds[..., 1], :] = out["coarse_is_background"]
    col = torch.ones([B, OR, 3]) * bg_color[None, ...]
    col[inds[..., 0], inds[..., 1], :] = out["coarse_raycolor"]
    opa = torch.zeros([B, OR, SR])
    opa[inds[..., 0], inds[..., 1], :] = out["coarse_point_opacity"]
    out.update(full_coarse_raycolor=col, full_coarse_is_background=isbg, full_coarse_mask=1 - isbg, full_coarse_point_opacity=opa)
    return out



def drop_patch_rays(patch_size, patch_num, drop_ratio):
    """point_aggregators.py:14-23: ray rows (of the patch_num*patch_size square batch) whose image feature is dropped."""
    flag = np.zeros((patch_size * patch_num, patch_size * patch_num))
    n = int(patch_num * patch_num * drop_ratio)
    row, col = n // patch_num, n % patch_num
    flag[0:row * patch_size, :] = 1
    flag[row * patch_size:row * patch_size + patch_size, 0:col * patch_size] = 1
    return np.where(flag.flatten() == 1)[0]


def shipped_loss(full_raycolor, ray_mask, conf_coefficient, gt, zero_epsilon, w_color=1.0, w_zero_one=1e-4, frame_weight=None):
    """The two loss terms the shipped ScanNet scripts enable (dev_scripts/w_scannet_etf/scene241.sh:146-151):
    `ray_masked_coarse_raycolor` MSE (models/base_rendering_model.py:1113-1118) and the zero-one regulariser on
    conf_coefficient (:1228-1240).  Returns (total, color, zero_one).  The shell's compute_losses adds a constant 1e-6 per colour
    loss item on top (:1198; no gradient) -- tests/golden/train_*.npz keeps its value as `loss_compute_losses`.  frame_weight: the item's scalar,
    applied to loss_total after the colour items and BEFORE the zero-one items are added (:1204-1205 vs :1228-1240)."""
    m3 = (ray_mask > 0)[..., None].expand(-1, -1, 3)
    mo = torch.masked_select(full_raycolor, m3).reshape(1, -1, 3)
    mg = torch.masked_select(gt, m3).reshape(1, -1, 3)
    lc = F.mse_loss(mo, mg)
    val = torch.clamp(conf_coefficient, zero_epsilon, 1 - zero_epsilon)
    lz = torch.mean(torch.log(val) + torch.log(1 - val))
    col = lc * w_color
    if frame_weight is not None:
        col = col * frame_weight
    return col + lz * w_zero_one, lc, lz


def train_step(xyz, emb, conf, pdir, color, sd, q, campos, camrotc2w, raydir_all, bg_color, c2w_nearest, campos_nearest,
               intrinsic_nearest, images_nearest, vsize, gt, zero_epsilon, drop_ray_rows, raydist_mode_unit=1, dtype=None, use_nearest=4,
               frame_weight=None, frame_weight_n=None, blur=None):
    """Forward in train mode + autograd of shipped_loss.  blur = (kernels [N,ks,ks], patch_num, patch_size): the blur-handling module between the
    render and the losses (models/mvs_points_volumetric_model.py:145-146: blur_update_output replaces output["coarse_raycolor"]).  frame_weight: the item's scalar on loss_total (models/base_rendering_model.py:1204-1205);
    frame_weight_n [1,V]: the per-view weights of the image-feature merge (models/aggregators/point_aggregators.py:1202-1203).  Returns (outputs, loss triple, grads dict) with grads keyed
    `neural_points.points_*` and `aggregator.<param>` like the reference's named parameters.
    dtype=torch.float64 re-runs the same graph in double precision (the query result q is kept): the yardstick for how much
    of a gradient difference is fp32 rounding noise."""
    if dtype is not None and dtype != torch.float32:
        c = lambda t: t.to(dtype) if isinstance(t, torch.Tensor) and t.is_floating_point() else t
        q = dict(q, sample_loc_w=torch.as_tensor(np.ascontiguousarray(q["sample_loc_w"])).to(dtype))
        old = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            return train_step(c(xyz), c(emb), c(conf), c(pdir), c(color), {k: c(v) for k, v in sd.items()}, q, c(campos), c(camrotc2w),
                              c(raydir_all), c(bg_color), c(c2w_nearest), c(campos_nearest), c(intrinsic_nearest), c(images_nearest),
                              vsize, c(gt), zero_epsilon, drop_ray_rows, raydist_mode_unit, use_nearest=use_nearest, frame_weight=frame_weight,
                              frame_weight_n=c(frame_weight_n), blur=None if blur is None else (c(blur[0]), blur[1], blur[2]))
        finally:
            torch.set_default_dtype(old)
    leaves = dict(emb=emb.clone().requires_grad_(True), conf=conf.clone().requires_grad_(True),
                  pdir=pdir.clone().requires_grad_(True), color=color.clone().requires_grad_(True))
    sdl = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = render(xyz, leaves["emb"], leaves["conf"], leaves["pdir"], leaves["color"], sdl, q, campos, camrotc2w, raydir_all,
                 bg_color, c2w_nearest, campos_nearest, intrinsic_nearest, images_nearest, vsize, raydist_mode_unit,
                 is_train=True, drop_ray_rows=drop_ray_rows, use_nearest=use_nearest, frame_weight_n=frame_weight_n)
    col = out["full_coarse_raycolor"]
    if blur is not None:
        col, out["blur_select"] = blur_update_output(col, gt, blur[0], blur[1], blur[2])
        out["blurred_raycolor"] = col
    loss, lc, lz = shipped_loss(col, out["ray_mask"], out["conf_coefficient"], gt, zero_epsilon, frame_weight=frame_weight)
    loss.backward()
    grads = {"neural_points.points_embeding": leaves["emb"].grad, "neural_points.points_conf": leaves["conf"].grad,
             "neural_points.points_dir": leaves["pdir"].grad, "neural_points.points_color": leaves["color"].grad}
    for k, v in sdl.items():
        if v.grad is not None:
            grads["aggregator." + k] = v.grad
    return out, (loss.item(), lc.item(), lz.item()), grads


def blur_update_output(color, gt, kernels, patch_num, patch_size):
    """models/base_rendering_model.py:677-745 (faster_version).  color, gt [1, S*S, 3]; kernels [N, ks, ks].
    Returns (new colours [1, S*S, 3], selected candidate per patch [patch_num^2], N = un-blurred)."""
    pn, ps = patch_num, patch_size
    S = pn * ps
    N, ks = kernels.shape[0], kernels.shape[-1]
    to_patches = lambda t: t.reshape(1, S, S, 3).permute(0, 3, 1, 2)[0].reshape(3, pn, ps, pn, ps).permute(1, 3, 0, 2, 4).reshape(pn * pn, 3, ps, ps)
    cp, gp = to_patches(color), to_patches(gt)
    w = kernels[:, None]                                               # [N,1,ks,ks]: F.conv2d = cross-correlation
    x = cp.reshape(pn * pn * 3, 1, ps, ps)
    m = F.conv2d(torch.ones_like(x), w, padding=ks // 2)
    b = torch.cat((F.conv2d(x, w, padding=ks // 2) / m, x), dim=1).reshape(pn * pn, 3, N + 1, ps, ps)
    diff = torch.sum(torch.abs(b - gp[:, :, None]), dim=(1, 3, 4))
    sel = torch.argmin(diff, dim=1)
    best = b[torch.arange(pn * pn), :, sel]                            # [P,3,ps,ps]
    out = best.reshape(pn, pn, 3, ps, ps).permute(2, 0, 3, 1, 4).reshape(3, S, S).permute(1, 2, 0).reshape(1, S * S, 3)
    return out, sel


def learnable_blur_update_output(color, gt, blur_predictor, patch_num, patch_size, kernel_size, kernel_norm, kernel_mode, boundary_mode, kernel_conv):
    """models/base_rendering_model.py:827-1020 (faster_version).  color, gt [1, S*S, 3]; returns (new colours, kernels [N,1,ks,ks])."""
    pn, ps, ks = patch_num, patch_size, kernel_size
    S, N = pn * ps, pn * pn
    to_patches = lambda t: t.reshape(1, S, S, 3).permute(0, 3, 1, 2)[0].reshape(3, pn, ps, pn, ps).permute(1, 3, 0, 2, 4).reshape(N, 3, ps, ps)
    cp, gp = to_patches(color), to_patches(gt)
    if kernel_conv:                                                                    # :886-889
        pred = blur_predictor[1](blur_predictor[0](torch.cat((gp.mean(dim=1, keepdim=True), cp.mean(dim=1, keepdim=True)), dim=1)).view(N, -1))
    else:                                                                              # :891-893
        pred = blur_predictor(torch.cat((gp.mean(dim=1).view(N, -1), cp.mean(dim=1).view(N, -1)), dim=-1))
    if kernel_norm == 0:                                                               # :897-901
        k = pred[:, 0:ks * ks].view(N, 1, ks, ks)
        k = k / k.sum(dim=(2, 3), keepdim=True)
    else:
        k = F.softmax(pred[:, 0:ks * ks], dim=-1).view(N, 1, ks, ks)
    if kernel_mode == 4:                                                               # :906-910
        w = pred[:, -1][..., None, None, None]
        ident = torch.zeros_like(k)
        ident[:, :, ks // 2, ks // 2] = 1.0
        k = w * k + (1 - w) * ident
        k = k / k.sum(dim=(2, 3), keepdim=True)
    x = cp.permute(1, 0, 2, 3)                                                         # [3, N, ps, ps], groups = N
    ones = torch.ones_like(x)
    if boundary_mode == 0:                                                             # :915-923
        b = F.conv2d(x, k, padding=ks // 2, groups=N) / (F.conv2d(ones, k, padding=ks // 2, groups=N) + 1e-10)
    elif boundary_mode == 1:
        b = F.conv2d(x, k, padding=ks // 2, groups=N) + (1 - F.conv2d(ones, k, padding=ks // 2, groups=N)) * x
    else:
        b = F.conv2d(x, k, padding=ks // 2, groups=N) + (1 - F.conv2d(ones, k.clone().detach(), padding=ks // 2, groups=N)) * x
    b = b.permute(1, 0, 2, 3)                                                          # [N, 3, ps, ps]
    out = b.reshape(pn, pn, 3, ps, ps).permute(2, 0, 3, 1, 4).reshape(3, S, S).permute(1, 2, 0).reshape(1, S * S, 3)
    return out, k
