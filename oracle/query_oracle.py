"""ctypes front-end of oracle/query_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

PARITY UNPINNED for the query stage (see the header of query_oracle.c): the reference's
query kernels cannot run in this image and the reference holds no golden vector for them.
`hyperparameters()` below IS pinned: tests/golden/query_hparams.json was produced by the
imported reference's `lighting_fast_querier.get_hyperparameters`
(/root/reference/models/neural_points/query_point_indices_worldcoords.py:46-77).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_query.so")
_lib = None


def build(force=False):
    """Compile the C oracle with gcc (seconds)."""
    src = os.path.join(_HERE, "query_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _SO


def _load():
    global _lib
    if _lib is not None:
        return _lib
    build()
    lib = ctypes.CDLL(_SO)
    c_fp = ctypes.POINTER(ctypes.c_float)
    c_ip = ctypes.POINTER(ctypes.c_int32)
    lib.oq_build.restype = ctypes.c_void_p
    lib.oq_build.argtypes = [c_fp, ctypes.c_int, c_fp, c_fp, c_ip, c_ip, ctypes.c_int, ctypes.c_int]
    lib.oq_free.argtypes = [ctypes.c_void_p]
    lib.oq_grid_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
    for name, rt in (("oq_coor_occ", ctypes.POINTER(ctypes.c_uint8)), ("oq_coor_2_occ", c_ip),
                     ("oq_occ_2_pnts", c_ip), ("oq_occ_numpnts", c_ip)):
        getattr(lib, name).restype = rt
        getattr(lib, name).argtypes = [ctypes.c_void_p]
    lib.oq_set_fma_d2.argtypes = [ctypes.c_int]
    lib.oq_set_fma_d2.restype = None
    lib.oq_query.restype = ctypes.c_int
    lib.oq_query.argtypes = [ctypes.c_void_p, c_fp, c_fp, c_fp, ctypes.c_int, c_fp, ctypes.c_int,
                             ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, c_ip,
                             c_ip, c_fp, ctypes.POINTER(ctypes.c_int8), ctypes.POINTER(ctypes.c_int64),
                             c_ip, c_fp, c_ip]
    _lib = lib
    return lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def hyperparameters(xyz, vsize, vscale, kernel_size, ranges, radius_limit_scale):
    """Restates get_hyperparameters (query_point_indices_worldcoords.py:46-77) with numpy.

    xyz [N,3] float32.  Returns dict(origin f32[3], cell f32[3], dims i32[3], ranges_np f32[6],
    radius2 f32) -- the values the reference uploads as d_coord_shift / d_voxel_size /
    d_grid_size and passes as radius_limit2 (:685).
    """
    xyz = np.asarray(xyz, dtype=np.float32)
    min_xyz = xyz.min(axis=0)                                           # :56 (fp32)
    max_xyz = xyz.max(axis=0)
    vscale_np = np.array(vscale, dtype=np.int32)                        # :57
    scaled_vsize_np = (np.asarray(list(vsize)) * vscale_np).astype(np.float32)   # :58 (f64 product -> f32)
    if ranges is not None:                                              # :59-63
        min_xyz = np.maximum(min_xyz, np.asarray(ranges[:3], dtype=np.float32))
        max_xyz = np.minimum(max_xyz, np.asarray(ranges[3:], dtype=np.float32))
    # :64-65  f32 array * python ints / 2 -> float64, cast to f32 by torch.as_tensor, fp32 subtract
    half = (scaled_vsize_np * np.asarray(list(kernel_size)) / 2).astype(np.float32)
    min_xyz = (min_xyz - half).astype(np.float32)
    max_xyz = (max_xyz + half).astype(np.float32)
    ranges_np = np.concatenate([min_xyz, max_xyz]).astype(np.float32)   # :67
    vdim_np = (max_xyz - min_xyz) / np.asarray(list(vsize))             # :69 (f32 diff / f64 list -> f64)
    scaled_vdim_np = np.ceil(vdim_np / vscale_np).astype(np.int32)      # :71
    radius_limit = np.asarray(radius_limit_scale * max(vsize[0], vsize[1])).astype(np.float32)   # :76-77
    radius2 = np.float32(radius_limit ** 2)                             # :685
    return dict(origin=ranges_np[:3].copy(), cell=scaled_vsize_np, dims=scaled_vdim_np,
                ranges_np=ranges_np, radius2=radius2)


class OracleGrid:
    """Serial restatement of build_occ_vox (:540-602)."""

    def __init__(self, xyz, origin, cell, dims, query_size, P, max_o):
        lib = _load()
        self.xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        self.origin = np.ascontiguousarray(origin, dtype=np.float32)
        self.cell = np.ascontiguousarray(cell, dtype=np.float32)
        self.dims = np.ascontiguousarray(dims, dtype=np.int32)
        self.query_size = np.ascontiguousarray(query_size, dtype=np.int32)
        self.P, self.max_o = int(P), int(max_o)
        self._h = lib.oq_build(_fp(self.xyz), self.xyz.shape[0], _fp(self.origin), _fp(self.cell),
                               _ip(self.dims), _ip(self.query_size), self.P, self.max_o)
        if not self._h:
            raise RuntimeError("oq_build failed (bad arguments or out of memory)")

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.oq_free(self._h)
            self._h = None

    def info(self):
        out = (ctypes.c_int64 * 8)()
        _lib.oq_grid_info(self._h, out)
        keys = ["n_occ", "n_inbounds", "n_dropped_voxels", "n_points_over_P", "n_cells_over_P",
                "vol", "n_dilated", "n_claimed"]
        return dict(zip(keys, [int(v) for v in out]))

    def tables(self):
        """(coor_occ u8[X,Y,Z], coor_2_occ i32[X,Y,Z], occ_2_pnts i32[max_o,P], occ_numpnts i32[max_o]) copies."""
        vol = int(np.prod(self.dims.astype(np.int64)))
        d = tuple(int(v) for v in self.dims)
        occ = np.ctypeslib.as_array(_lib.oq_coor_occ(self._h), shape=(vol,)).reshape(d).copy()
        c2o = np.ctypeslib.as_array(_lib.oq_coor_2_occ(self._h), shape=(vol,)).reshape(d).copy()
        o2p = np.ctypeslib.as_array(_lib.oq_occ_2_pnts(self._h), shape=(self.max_o * self.P,)).reshape(self.max_o, self.P).copy()
        onp = np.ctypeslib.as_array(_lib.oq_occ_numpnts(self._h), shape=(self.max_o,)).copy()
        return occ, c2o, o2p, onp

    def query(self, campos, raydir, tmid, SR, K, radius2, kernel_size, want_full=False, fma_d2=False):
        """Restates query_grid_point_index (:605-711) after build_occ_vox.
        fma_d2: evaluate the candidate distance (:492) with the FMA chain nvcc -fmad=true emits (see query_oracle.c header).

        Returns dict(sample_pidx [R',SR,K] i32, sample_loc_w [R',SR,3] f32, ray_mask [R] i8,
        counts dict[, full_pidx, full_loc, full_nsamp]).
        """
        campos = np.ascontiguousarray(campos, dtype=np.float32).reshape(3)
        raydir = np.ascontiguousarray(raydir, dtype=np.float32).reshape(-1, 3)
        tmid = np.ascontiguousarray(tmid, dtype=np.float32)
        R = raydir.shape[0]
        if tmid.ndim == 1:
            D, stride = tmid.shape[0], 0
        else:
            assert tmid.shape[0] == R
            D, stride = tmid.shape[1], tmid.shape[1]
        ks = np.ascontiguousarray(kernel_size, dtype=np.int32)
        pidx = np.empty((R, SR, K), dtype=np.int32)
        loc = np.empty((R, SR, 3), dtype=np.float32)
        mask = np.zeros((R,), dtype=np.int8)
        counts = (ctypes.c_int64 * 8)()
        fp = fl = fn = None
        if want_full:
            fp = np.empty((R, SR, K), dtype=np.int32)
            fl = np.empty((R, SR, 3), dtype=np.float32)
            fn = np.empty((R,), dtype=np.int32)
        _lib.oq_set_fma_d2(1 if fma_d2 else 0)
        rc = _lib.oq_query(self._h, _fp(self.xyz), _fp(campos), _fp(raydir), R, _fp(tmid), D, stride,
                           int(SR), int(K), ctypes.c_float(float(radius2)), _ip(ks),
                           _ip(pidx), _fp(loc), mask.ctypes.data_as(ctypes.POINTER(ctypes.c_int8)), counts,
                           _ip(fp) if want_full else None, _fp(fl) if want_full else None,
                           _ip(fn) if want_full else None)
        _lib.oq_set_fma_d2(0)
        if rc != 0:
            raise RuntimeError("oq_query failed rc=%d" % rc)
        keys = ["n_valid_rays", "n_hit_rays", "n_samples", "n_neighbours", "n_cells_visited",
                "n_candidates", "n_valid_samples"]
        cd = dict(zip(keys, [int(v) for v in counts][:7]))
        nv = cd["n_valid_rays"]
        out = dict(sample_pidx=pidx[:nv].copy(), sample_loc_w=loc[:nv].copy(), ray_mask=mask, counts=cd)
        if want_full:
            out.update(full_pidx=fp, full_loc=fl, full_nsamp=fn)
        return out


def tmid_table(near, far, D):
    """t_mid for jitter=0, with the torch ops of near_far_linear_ray_generation
    (/root/reference/models/rendering/diff_ray_marching.py:369-385) on CPU, fp32."""
    import torch
    tvals = torch.linspace(0, 1, D + 1).view(1, -1)
    tvals = near * (1 - tvals) + far * tvals
    seg = (tvals[..., 1:] - tvals[..., :-1]) * (1 + 0.0 * (torch.zeros((1, 1, D)) - 0.5))
    end = torch.cumsum(seg, dim=2)
    end = torch.cat([torch.zeros((1, 1, 1)), end], dim=2)
    end = near + end
    mid = (end[:, :, :-1] + end[:, :, 1:]) / 2
    return mid.reshape(D).numpy().astype(np.float32)
