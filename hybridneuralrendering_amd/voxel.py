"""Voxel down-sampling of the initial point cloud on the device (SURVEY 8f "next" row 4).

Mirror of models/mvs/mvs_utils.py:537-563 `construct_vox_points_closest` (run/train_ft.py:164, :725), which needs torch_scatter:
same signature for the shipped call form `(xyz_val, vox_res)`, same return triple `(xyz_centroid, sparse_grid_idx, min_idx)`.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import HnrError


def space_of(xyz_min, xyz_max, vox_res):
    """(space_min [3] f32, vox_size f32) exactly as the reference's fp32 tensor ops compute them (:541-549)."""
    mn, mx = np.asarray(xyz_min, np.float32), np.asarray(xyz_max, np.float32)
    edge = np.float32(np.max(mx - mn) * np.float32(1.05))
    mid = (mx + mn) / np.float32(2)
    space_min = (mid - edge / np.float32(2)).astype(np.float32)
    return space_min, np.float32(edge / np.float32(vox_res))


def construct_vox_points_closest(xyz_val, vox_res, partition_xyz=None, space_min=None, space_max=None, return_inverse=False):
    """xyz_val [N,3] fp32 on the GPU -> (xyz_centroid [V,3], sparse_grid_idx [V,3] int32 in torch.unique order, min_idx [V] int64)."""
    if partition_xyz is not None or space_min is not None or space_max is not None:
        raise HnrError("construct_vox_points_closest: only the shipped call form (xyz, vox_res) is implemented")
    L = _lib.lib()
    xyz = _lib.require_gpu(xyz_val, "xyz_val", torch.float32)
    if xyz.dim() != 2 or xyz.shape[1] != 3:
        raise HnrError("xyz_val must be [N,3]")
    n, dev = int(xyz.shape[0]), xyz.device
    if n == 0:
        return xyz.new_zeros((0, 3)), torch.zeros((0, 3), dtype=torch.int32, device=dev), torch.zeros((0,), dtype=torch.int64, device=dev)
    b6 = torch.empty((6,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.hnr_points_bounds(_lib.ptr(xyz), n, _lib.ptr(b6), _lib.stream()), "hnr_points_bounds")
    b = b6.cpu().numpy()
    smin, vsz = space_of(b[:3], b[3:], vox_res)
    cen = torch.empty((n, 3), dtype=torch.float32, device=dev)
    gidx = torch.empty((n, 3), dtype=torch.int32, device=dev)
    midx = torch.empty((n,), dtype=torch.int32, device=dev)
    inv = torch.empty((n,), dtype=torch.int32, device=dev) if return_inverse else None
    cnt = torch.zeros((1,), dtype=torch.int64, device=dev)
    nbytes = int(L.hnr_voxel_downsample_scratch_bytes(n))
    scratch = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    sm = (ctypes.c_float * 3)(*[float(v) for v in smin])
    with torch.cuda.device(dev):
        _lib.check(L.hnr_voxel_downsample(_lib.ptr(xyz), n, sm, float(vsz), _lib.ptr(cen), _lib.ptr(gidx), _lib.ptr(midx),
                                          _lib.ptr(inv) if inv is not None else None, _lib.ptr(cnt), _lib.ptr(scratch), nbytes, _lib.stream()),
                   "hnr_voxel_downsample")
    v = int(cnt.item())
    out = (cen[:v], gidx[:v], midx[:v].long())
    return out + (inv.long(),) if return_inverse else out
