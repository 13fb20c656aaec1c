"""CPU restatement (TEST INFRASTRUCTURE ONLY: imported by tests/ and nothing else) of the reference's voxel down-sampling,
models/mvs/mvs_utils.py:537-563 `construct_vox_points_closest`, for the shipped call form (xyz, vox_res).

Pinning: the reference function itself needs torch_scatter, which this image lacks.  tests/golden/make_golden.py::gen_voxel runs the
reference function with a stand-in for its two torch_scatter calls (scatter_mean = index_add / count, scatter_min = first
minimum) -- everything else (bounds, fp32 cell arithmetic, torch.unique order, residual norm) is the reference's own code --
and this file is checked against that fixture (tests/golden/voxel_down.npz).  The stand-ins themselves are pinned to the DEFINITION of
the two operators by brute force in float64 (tests/test_voxel.py::test_golden_scatter_outputs_satisfy_the_definition_of_torch_scatter).
The two scatter calls are therefore restated,
not pinned: torch_scatter's CUDA path sums with atomics and resolves argmin ties by race, so no fixture could pin them anyway.
"""
import numpy as np


def construct_vox_points_closest(xyz, vox_res):
    xyz = np.asarray(xyz, np.float32)
    mn, mx = xyz.min(axis=0), xyz.max(axis=0)                                   # :541
    edge = np.float32(np.max(mx - mn) * np.float32(1.05))                       # :542
    mid = (mx + mn) / np.float32(2)                                             # :543
    space_min = (mid - edge / np.float32(2)).astype(np.float32)                 # :544
    sz = np.float32(edge / np.float32(vox_res))                                 # :551
    cell = np.floor((xyz - space_min[None]) / sz).astype(np.int32)              # :552-553
    grid, inv = np.unique(cell, axis=0, return_inverse=True)                    # lexicographic rows, like torch.unique(dim=0)
    inv = inv.reshape(-1)
    V = grid.shape[0]
    s = np.zeros((V, 3), np.float32)
    np.add.at(s, inv, xyz)                                                      # sequential fp32 sums in point order (:554)
    cnt = np.bincount(inv, minlength=V).astype(np.float32)
    cen = (s / cnt[:, None]).astype(np.float32)
    d = xyz - cen[inv]
    res = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).astype(np.float32)   # :556
    order = np.lexsort((np.arange(len(inv)), res, inv))                         # per voxel: smallest residual, then smallest id (:559)
    first = np.r_[True, inv[order][1:] != inv[order][:-1]]
    min_idx = order[first].astype(np.int64)
    return cen, grid.astype(np.int32), min_idx, inv.astype(np.int64), res
