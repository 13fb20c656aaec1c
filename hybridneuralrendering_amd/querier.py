"""Host-side mirror of the reference's world-coordinate querier over libhnr_hip.so.

Drop-in for `models.neural_points.query_point_indices_worldcoords.lighting_fast_querier`
(/root/reference/models/neural_points/query_point_indices_worldcoords.py:29-93): same constructor,
same `query_points` signature and 7-tuple, same `get_hyperparameters` arithmetic, `clean_up()`.

Differences that do not change results:
  * the voxel grid is built once per point-cloud version (keyed on storage pointer, shape, tensor
    version and the grid-defining options) instead of once per call;
  * no pycuda context, no JIT: kernels are ahead-of-time gfx950 code in libhnr_hip.so;
  * one host read (the valid-ray count that sizes the returned tensors) instead of two.
Where the reference is nondeterministic (atomics order, wall-clock-seeded reservoir) this follows
the serial linearisation of oracle/query_oracle.c and REPORTS overflow (`last_grid_stats`).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import HnrError, GridParams, GridStats, QueryParams, CNT, NCOUNTS


def tmid_table(near, far, D, device=None):
    """Marched depths for jitter = 0: the op chain of near_far_linear_ray_generation
    (/root/reference/models/rendering/diff_ray_marching.py:369-385) evaluated once on the host in
    fp32; identical for every ray, so it is a [D] kernel input instead of a [R,D,3] tensor."""
    tvals = torch.linspace(0, 1, D + 1).view(1, -1)
    tvals = near * (1 - tvals) + far * tvals
    seg = (tvals[..., 1:] - tvals[..., :-1]).view(1, 1, D) * (1 + 0.0 * (torch.zeros((1, 1, D)) - 0.5))
    end = torch.cumsum(seg, dim=2)
    end = torch.cat([torch.zeros((1, 1, 1)), end], dim=2)
    end = near + end
    mid = ((end[:, :, :-1] + end[:, :, 1:]) / 2).reshape(D).contiguous()
    return mid.to(device) if device is not None else mid


def tmid_jittered(near, far, D, R, jitter, device, generator=None):
    """Per-ray marched depths for train-time jitter (same op chain, `torch.rand` on the device,
    diff_ray_marching.py:372-385).  Returns [R, D]."""
    tvals = torch.linspace(0, 1, D + 1, device=device).view(1, -1)
    tvals = near * (1 - tvals) + far * tvals
    seg = (tvals[..., 1:] - tvals[..., :-1]) * (1 + jitter * (torch.rand((1, R, D), device=device, generator=generator) - 0.5))
    end = torch.cumsum(seg, dim=2)
    end = torch.cat([torch.zeros((1, R, 1), device=device), end], dim=2)
    end = near + end
    return ((end[:, :, :-1] + end[:, :, 1:]) / 2).reshape(R, D).contiguous()


def compute_hyperparameters(min_xyz, max_xyz, vsize, vscale, kernel_size, ranges, radius_limit_scale):
    """The arithmetic of get_hyperparameters (:56-77) given the fp32 bounds of the cloud (numpy f32[3] each).
    Returns (radius_limit f32, ranges_np f32[6], scaled_vsize_np f32[3], scaled_vdim_np i32[3], vdim_np f64[3])."""
    min_xyz = np.asarray(min_xyz, dtype=np.float32)
    max_xyz = np.asarray(max_xyz, dtype=np.float32)
    vsize_np = np.asarray(list(vsize))                                  # python floats -> f64 (list * ndarray)
    vscale_np = np.array(vscale, dtype=np.int32)
    scaled_vsize_np = (vsize_np * vscale_np).astype(np.float32)         # :58
    if ranges is not None:                                              # :63
        min_xyz = np.maximum(min_xyz, np.asarray(ranges[:3], dtype=np.float32))
        max_xyz = np.minimum(max_xyz, np.asarray(ranges[3:], dtype=np.float32))
    pad = (scaled_vsize_np * np.asarray(list(kernel_size)) / 2).astype(np.float32)   # :64 f64 -> as_tensor(f32)
    min_xyz = (min_xyz - pad).astype(np.float32)
    max_xyz = (max_xyz + pad).astype(np.float32)
    ranges_np = np.concatenate([min_xyz, max_xyz]).astype(np.float32)   # :67
    vdim_np = (max_xyz - min_xyz) / vsize_np                            # :69
    scaled_vdim_np = np.ceil(vdim_np / vscale_np).astype(np.int32)      # :71
    radius_limit_np = np.asarray(radius_limit_scale * max(vsize[0], vsize[1])).astype(np.float32)   # :76-77
    return radius_limit_np, ranges_np, scaled_vsize_np, scaled_vdim_np, vdim_np


class VoxelGrid:
    """Owns an hnr_grid handle (device index arrays) for one version of the point cloud."""

    def __init__(self, xyz, origin, cell, dims, query_size, P, max_o):
        L = _lib.lib()
        xyz = _lib.require_gpu(xyz, "xyz", torch.float32).reshape(-1, 3)
        self.device = xyz.device
        self.n_points = xyz.shape[0]
        p = GridParams()
        for a in range(3):
            p.origin[a] = float(origin[a]); p.cell[a] = float(cell[a])
            p.dims[a] = int(dims[a]); p.query_size[a] = int(query_size[a])
        p.P, p.max_o = int(P), int(max_o)
        self.params = p
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(L.hnr_grid_build(_lib.ptr(xyz), self.n_points, ctypes.byref(p), _lib.stream(), ctypes.byref(h)),
                       "hnr_grid_build")
        self._h = h
        st = GridStats()
        _lib.check(L.hnr_grid_get_stats(self._h, ctypes.byref(st)), "hnr_grid_get_stats")
        self.stats = {k: int(getattr(st, k)) for k, _ in GridStats._fields_}

    @property
    def handle(self):
        if self._h is None:
            raise HnrError("VoxelGrid used after free()")
        return self._h

    def free(self):
        if getattr(self, "_h", None) is not None:
            _lib.lib().hnr_grid_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def grow(self, xyz):
        """hnr_grid_grow: extend the tables in place for a cloud whose first `n_points` rows are the points this grid was built from and whose
        other rows were appended (NeuralPoints.grow_points).  True: done; False: the library cannot do it in place (slack used up, max_o, no
        neighbourhood lists) -- nothing changed, rebuild."""
        xyz = _lib.require_gpu(xyz, "xyz", torch.float32).reshape(-1, 3)
        if xyz.shape[0] < self.n_points:
            raise HnrError("VoxelGrid.grow: %d points, the grid describes %d (points can only be appended)" % (xyz.shape[0], self.n_points))
        with torch.cuda.device(self.device):
            rc = _lib.lib().hnr_grid_grow(self.handle, _lib.ptr(xyz), int(xyz.shape[0]), _lib.stream())
        if rc == 1:                                               # HNR_NEED_REBUILD
            return False
        _lib.check(rc, "hnr_grid_grow")
        self.n_points = int(xyz.shape[0])
        st = GridStats()
        _lib.check(_lib.lib().hnr_grid_get_stats(self._h, ctypes.byref(st)), "hnr_grid_get_stats")
        self.stats = {k: int(getattr(st, k)) for k, _ in GridStats._fields_}
        return True

    def export_runs(self):
        """(run_len i32[X,Y,Z] (-1 outside the dilated mask), run_hash i64[X,Y,Z]) of the 3x3x3 neighbourhood lists -- test hook."""
        d = tuple(int(self.params.dims[a]) for a in range(3))
        ln = torch.empty(d, dtype=torch.int32, device=self.device)
        hs = torch.empty(d, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().hnr_grid_export_runs(self.handle, _lib.ptr(ln), _lib.ptr(hs), _lib.stream()), "hnr_grid_export_runs")
        return ln, hs

    def export_dense(self):
        """(coor_occ u8[X,Y,Z], cell_count i32[X,Y,Z], cell_first i32[X,Y,Z]) -- test hook."""
        d = tuple(int(self.params.dims[a]) for a in range(3))
        occ = torch.empty(d, dtype=torch.uint8, device=self.device)
        cnt = torch.empty(d, dtype=torch.int32, device=self.device)
        first = torch.empty(d, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().hnr_grid_export_dense(self.handle, _lib.ptr(occ), _lib.ptr(cnt), _lib.ptr(first),
                                                        _lib.stream()), "hnr_grid_export_dense")
        return occ, cnt, first


def points_bounds(xyz):
    """fp32 min/max over the cloud on the device -> (min f32[3], max f32[3]) numpy (one host read)."""
    xyz = _lib.require_gpu(xyz, "xyz", torch.float32).reshape(-1, 3)
    out = torch.empty(6, dtype=torch.float32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        _lib.check(_lib.lib().hnr_points_bounds(_lib.ptr(xyz), xyz.shape[0], _lib.ptr(out), _lib.stream()),
                   "hnr_points_bounds")
    o = out.cpu().numpy()
    return o[:3].copy(), o[3:].copy()


def march_query(grid, campos, raydir, tmid, SR, K, radius2, kernel_size, pad=True, knn_order=0):
    """hnr_march_query on R rays.  Un-compacted outputs (row r = input ray r), no host sync:
    dict(sample_pidx [R,SR,K] i32, sample_loc_w [R,SR,3] f32, ray_nsamp [R] i32, ray_mask [R] i8,
    counts [8] i64 (device)).  knn_order = 1: the neighbour SET of the reference rule in canonical order (ascending distance), K = 8 only."""
    L = _lib.lib()
    campos = _lib.require_gpu(campos, "campos", torch.float32).reshape(3)
    raydir = _lib.require_gpu(raydir, "raydir", torch.float32).reshape(-1, 3)
    tmid = _lib.require_gpu(tmid, "tmid", torch.float32)
    dev = raydir.device
    R = raydir.shape[0]
    q = QueryParams()
    q.R, q.SR, q.K = R, int(SR), int(K)
    if tmid.dim() == 1:
        q.D, q.tmid_stride = tmid.shape[0], 0
    else:
        if tmid.shape[0] != R:
            raise HnrError("per-ray tmid must be [R, D]")
        q.D, q.tmid_stride = tmid.shape[1], tmid.shape[1]
    for a in range(3):
        q.kernel_size[a] = int(kernel_size[a])
    q.radius2 = float(radius2)
    q.pad_outputs = 1 if pad else 0
    q.knn_order = int(knn_order)
    pidx = torch.empty((R, SR, K), dtype=torch.int32, device=dev)
    loc = torch.empty((R, SR, 3), dtype=torch.float32, device=dev)
    nsamp = torch.empty((R,), dtype=torch.int32, device=dev)
    mask = torch.empty((R,), dtype=torch.int8, device=dev)
    work = torch.empty((max(int(L.hnr_query_work_elems(R, int(SR))), 2),), dtype=torch.int32, device=dev)
    counts = torch.empty((NCOUNTS,), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.hnr_march_query(grid.handle, _lib.ptr(campos), _lib.ptr(raydir), _lib.ptr(tmid), ctypes.byref(q),
                                     _lib.ptr(pidx), _lib.ptr(loc), _lib.ptr(nsamp), _lib.ptr(mask), _lib.ptr(work),
                                     _lib.ptr(counts), _lib.stream()), "hnr_march_query")
    return dict(sample_pidx=pidx, sample_loc_w=loc, ray_nsamp=nsamp, ray_mask=mask, counts=counts, work=work, padded=bool(pad))


def compact_rays(res, raydir, campos, camrot):
    """Second compaction (:705-709) + ray-dir expansion (:91) + w2pers (:96-103).
    Reads the valid-ray count on the host (the returned tensors are sized by it)."""
    L = _lib.lib()
    pidx, loc, mask, counts = res["sample_pidx"], res["sample_loc_w"], res["ray_mask"], res["counts"]
    R, SR, K = pidx.shape
    dev = pidx.device
    raydir = _lib.require_gpu(raydir, "raydir", torch.float32).reshape(-1, 3)
    campos = _lib.require_gpu(campos, "campos", torch.float32).reshape(3)
    camrot = _lib.require_gpu(camrot, "camrot", torch.float32).reshape(3, 3)
    row = torch.empty((max(R, 1),), dtype=torch.int32, device=dev)
    scratch = torch.empty(((R + 1023) // 1024 + 1,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.hnr_ray_compact_plan(_lib.ptr(mask), R, _lib.ptr(row), _lib.ptr(scratch), _lib.ptr(counts),
                                          _lib.stream()), "hnr_ray_compact_plan")
        n_valid = int(counts[CNT["RAYS_VALID"]].item())           # the one host read
        o_pidx = torch.empty((n_valid, SR, K), dtype=torch.int32, device=dev)
        o_loc = torch.empty((n_valid, SR, 3), dtype=torch.float32, device=dev)
        o_pers = torch.empty((n_valid, SR, 3), dtype=torch.float32, device=dev)
        o_dir = torch.empty((n_valid, SR, 3), dtype=torch.float32, device=dev)
        if n_valid > 0:
            _lib.check(L.hnr_ray_compact(_lib.ptr(row), R, SR, K, _lib.ptr(pidx), _lib.ptr(loc), _lib.ptr(raydir),
                                         _lib.ptr(campos), _lib.ptr(camrot), _lib.ptr(o_pidx), _lib.ptr(o_loc),
                                         _lib.ptr(o_pers), _lib.ptr(o_dir), _lib.stream()), "hnr_ray_compact")
    return o_pidx, o_pers, o_loc, o_dir, row[:R]


class lighting_fast_querier:
    """Same surface as the reference class (query_point_indices_worldcoords.py:29-93)."""

    def __init__(self, device, opt):
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        if self.device.type != "cuda":
            raise HnrError("lighting_fast_querier needs a GPU device; there is no CPU path")
        self.gpu = self.device.index
        self.opt = opt
        if getattr(opt, "NN", 2) <= 0:
            # build_cuda asks for a kernel that does not exist when NN == 0 (:530)
            raise HnrError("opt.NN must be > 0 (the world-coord querier has no query_rand_along_ray)")
        if getattr(opt, "inverse", 0) > 0:
            raise HnrError("opt.inverse > 0 (disparity-linear marching) is not used by any shipped config and is unsupported")
        self.inverse = getattr(opt, "inverse", 0)
        self.count = 0
        self._grid = None
        self._grid_key = None
        self._grid_src = None
        self._hp = None
        self._tmid = {}
        self.last_counts = None
        self.last_grid_stats = None
        _lib.lib()   # fail now, loudly, if the HIP library is absent

    def clean_up(self):
        if self._grid is not None:
            self._grid.free()
        self._grid, self._grid_key, self._hp, self._grid_src, self._grid_bounds = None, None, None, None, None

    # -- hyper-parameters (:46-77) -------------------------------------------------------------
    def get_hyperparameters(self, vsize_np, point_xyz_w_tensor, ranges=None, bounds=None):
        """bounds: (min, max) of the cloud when the caller knows them already (grow: old bounds combined with the new points')."""
        mn, mx = points_bounds(point_xyz_w_tensor) if bounds is None else bounds
        self._last_bounds = (np.asarray(mn, np.float32).copy(), np.asarray(mx, np.float32).copy())
        radius_limit_np, ranges_np, scaled_vsize_np, scaled_vdim_np, vdim_np = compute_hyperparameters(
            mn, mx, vsize_np, self.opt.vscale, self.opt.kernel_size, ranges, self.opt.radius_limit_scale)
        depth_limit_np = np.asarray(getattr(self.opt, "depth_limit_scale", 0.0) * vsize_np[2]).astype(np.float32)
        vscale_np = np.array(self.opt.vscale, dtype=np.int32)
        dev = point_xyz_w_tensor.device
        to_dev = lambda a: torch.as_tensor(a, device=dev)
        return (radius_limit_np, depth_limit_np, ranges_np, vsize_np, vdim_np, scaled_vsize_np, scaled_vdim_np, vscale_np,
                to_dev(ranges_np), to_dev(scaled_vsize_np), to_dev(scaled_vdim_np), to_dev(vscale_np),
                to_dev(np.asarray(self.opt.kernel_size, dtype=np.int32)), to_dev(np.asarray(self.opt.query_size, dtype=np.int32)))

    def _grid_for(self, point_xyz_w_tensor):
        xyz = point_xyz_w_tensor.detach()
        key = self._key_of(xyz)
        if self._grid is not None and key == self._grid_key:
            return self._grid, self._hp
        self.clean_up()
        hp = self.get_hyperparameters(self.opt.vsize, point_xyz_w_tensor, ranges=self.opt.ranges)
        self._grid_bounds = self._last_bounds
        radius_limit_np, _, ranges_np, _, _, scaled_vsize_np, scaled_vdim_np = hp[:7]
        self._grid = VoxelGrid(xyz.reshape(-1, 3), ranges_np[:3], scaled_vsize_np, scaled_vdim_np, self.opt.query_size,
                               self.opt.P, self.opt.max_o)
        self._grid_key, self._hp = key, hp
        self._grid_src = xyz                                     # keeps the keyed buffer alive (its address must not be recycled)
        self.last_grid_stats = self._grid.stats
        return self._grid, hp

    def _key_of(self, xyz):
        return (xyz.data_ptr(), tuple(xyz.shape), xyz._version, tuple(self.opt.vsize), tuple(self.opt.vscale),
                tuple(self.opt.kernel_size), tuple(self.opt.query_size), tuple(self.opt.ranges) if self.opt.ranges is not None else None,
                int(self.opt.P), int(self.opt.max_o), float(self.opt.radius_limit_scale))

    def grow(self, point_xyz_w_tensor, n_old):
        """After NeuralPoints.grow_points (neural_points.py:376-402): `point_xyz_w_tensor` [1,N,3] / [N,3] is the grown cloud whose first n_old rows
        are the cloud the cached grid was built from.  When the grown cloud gives the SAME grid geometry (get_hyperparameters: its bounding box
        decides origin and dims) the tables are extended in place (hnr_grid_grow: 0.85 - 0.9 ms for 1 % new points at 2 M against a 4.7 ms rebuild) and
        True is returned; otherwise the cache is dropped and the next query rebuilds.  Logically identical to a rebuild either way."""
        xyz = point_xyz_w_tensor.detach()
        if self._grid is None or self._hp is None or self._grid.n_points != int(n_old):
            self.clean_up()
            return False
        # the grown cloud's bounds = the old cloud's combined with the NEW points' (min / max are exact: the same values as a pass over all N points)
        old_b = getattr(self, "_grid_bounds", None)                    # bounds of the cloud the cached grid describes (_grid_for / grow)
        flat = xyz.reshape(-1, 3)
        if old_b is None or flat.shape[0] <= int(n_old):
            self.clean_up()
            return False
        mn2, mx2 = points_bounds(flat[int(n_old):])
        hp = self.get_hyperparameters(self.opt.vsize, xyz if xyz.dim() == 3 else xyz[None], ranges=self.opt.ranges,
                                      bounds=(np.minimum(old_b[0], mn2), np.maximum(old_b[1], mx2)))
        same = all(np.array_equal(np.asarray(hp[i]), np.asarray(self._hp[i])) for i in (0, 2, 5, 6))
        if not same or not self._grid.grow(xyz.reshape(-1, 3)):
            self.clean_up()
            return False
        self._grid_key, self._hp, self._grid_src = self._key_of(xyz if xyz.dim() == 3 else xyz[None]), hp, xyz
        self._grid_bounds = self._last_bounds
        self.last_grid_stats = self._grid.stats
        return True

    def _tmid_for(self, near, far, D, R, device):
        if getattr(self.opt, "is_train", 0) > 0:
            return tmid_jittered(near, far, D, R, 0.3, device)          # :87 jitter=0.3 when training
        key = (float(near), float(far), int(D), str(device))
        if key not in self._tmid:
            self._tmid[key] = tmid_table(near, far, D, device=device)
        return self._tmid[key]

    # -- query (:80-93) ------------------------------------------------------------------------
    def query_points(self, pixel_idx_tensor, point_xyz_pers_tensor, point_xyz_w_tensor, actual_numpoints_tensor, h, w,
                     intrinsic, near_depth, far_depth, ray_dirs_tensor, cam_pos_tensor, cam_rot_tensor):
        near_depth, far_depth = np.asarray(near_depth).item(), np.asarray(far_depth).item()
        if point_xyz_w_tensor.shape[0] != 1:
            raise HnrError("batch size must be 1 (it always is in the reference)")
        grid, hp = self._grid_for(point_xyz_w_tensor)
        radius_limit_np, _, ranges_np = hp[0], hp[1], hp[2]
        rays = ray_dirs_tensor.reshape(-1, 3)
        tmid = self._tmid_for(near_depth, far_depth, self.opt.z_depth_dim, rays.shape[0], rays.device)
        res = march_query(grid, cam_pos_tensor.reshape(3), rays, tmid, self.opt.SR, self.opt.K,
                          np.float32(radius_limit_np ** 2), self.opt.kernel_size)
        pidx, loc_pers, loc_w, dirs, _ = compact_rays(res, rays, cam_pos_tensor.reshape(3), cam_rot_tensor.reshape(3, 3))
        self.last_counts = res["counts"]
        ray_mask = res["ray_mask"][None, :]
        return pidx[None], loc_pers[None], loc_w[None], dirs[None], ray_mask, self.opt.vsize, ranges_np

    def w2pers(self, point_xyz_w, camrotc2w, campos):
        # :96-103 (kept for API completeness; query_points computes it inside hnr_ray_compact)
        xyz_w_shift = point_xyz_w - campos[:, None, :]
        xyz_c = torch.sum(xyz_w_shift[..., None, :] * torch.transpose(camrotc2w, 1, 2)[:, None, None, ...], dim=-1)
        return torch.stack([xyz_c[..., 0] / xyz_c[..., 2], xyz_c[..., 1] / xyz_c[..., 2], xyz_c[..., 2]], dim=-1)
