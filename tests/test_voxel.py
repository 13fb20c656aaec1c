"""Voxel down-sampling (models/mvs/mvs_utils.py:537-563 construct_vox_points_closest): oracle vs the reference-generated golden
(CPU) and the HIP path vs both (GPU).  Cells, voxel order and centroids must agree exactly / to fp32 rounding; the chosen point
per voxel may differ only where two residuals are within fp32 rounding of each other (torch.norm's reduction order is not ours)."""
import os

import numpy as np
import pytest
import torch

from tests.golden_io import GOLD


def _check_choice(xyz, cen, inv, got_idx, exp_idx):
    diff = np.nonzero(got_idx != exp_idx)[0]
    for v in diff:                                   # a different pick must be a tie to fp32 rounding, and inside the same voxel
        a, b = got_idx[v], exp_idx[v]
        assert inv[a] == v and inv[b] == v
        ra, rb = np.linalg.norm(xyz[a].astype(np.float64) - cen[v]), np.linalg.norm(xyz[b].astype(np.float64) - cen[v])
        assert abs(ra - rb) <= 1e-6 * max(ra, rb, 1e-6), (v, ra, rb)
    return len(diff)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_voxel_oracle_matches_reference_golden(tag):
    from oracle import voxel_oracle as vo
    z = np.load(os.path.join(GOLD, "voxel_down.npz"))
    xyz, res = z[tag + "_xyz"], float(z[tag + "_res"][0])
    cen, grid, midx, inv, _ = vo.construct_vox_points_closest(xyz, res)
    np.testing.assert_array_equal(grid, z[tag + "_grid"])
    np.testing.assert_allclose(cen, z[tag + "_centroid"], rtol=0, atol=1e-6)
    assert _check_choice(xyz, cen, inv, midx, z[tag + "_min_idx"]) <= 0.002 * len(midx)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_voxel_hip_matches_golden_and_oracle(tag):
    from hybridneuralrendering_amd.voxel import construct_vox_points_closest
    from oracle import voxel_oracle as vo
    z = np.load(os.path.join(GOLD, "voxel_down.npz"))
    xyz, res = z[tag + "_xyz"], float(z[tag + "_res"][0])
    cen, grid, midx, inv = construct_vox_points_closest(torch.from_numpy(xyz).cuda(), res, return_inverse=True)
    ocen, ogrid, omidx, oinv, _ = vo.construct_vox_points_closest(xyz, res)
    np.testing.assert_array_equal(grid.cpu().numpy(), z[tag + "_grid"])
    np.testing.assert_array_equal(grid.cpu().numpy(), ogrid)
    np.testing.assert_array_equal(inv.cpu().numpy(), oinv)
    np.testing.assert_array_equal(cen.cpu().numpy(), ocen)                      # same sequential fp32 sums -> bit-exact vs the oracle
    np.testing.assert_allclose(cen.cpu().numpy(), z[tag + "_centroid"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(midx.cpu().numpy(), omidx)
    assert _check_choice(xyz, ocen, oinv, midx.cpu().numpy(), z[tag + "_min_idx"]) <= 0.002 * len(omidx)


@pytest.mark.gpu
def test_voxel_hip_large_cloud_properties_and_edge_cases():
    """2 M points (the bench cloud size): every voxel's representative lies in the voxel, voxels are unique and sorted, counts add up."""
    from hybridneuralrendering_amd.voxel import construct_vox_points_closest, space_of
    from hybridneuralrendering_amd._lib import HnrError
    g = torch.Generator().manual_seed(3)
    xyz = (torch.rand((2000000, 3), generator=g) * torch.tensor([8.0, 6.0, 3.0])).cuda()
    cen, grid, midx, inv = construct_vox_points_closest(xyz, 320, return_inverse=True)
    V = grid.shape[0]
    key = (grid[:, 0].long() * 4096 + grid[:, 1].long()) * 4096 + grid[:, 2].long()
    assert torch.all(key[1:] > key[:-1])                                         # unique, lexicographic
    assert int(inv.max()) == V - 1 and torch.equal(inv[midx], torch.arange(V, device="cuda"))
    smin, vsz = space_of(xyz.min(dim=0)[0].cpu().numpy(), xyz.max(dim=0)[0].cpu().numpy(), 320)
    cell = torch.floor((xyz - torch.from_numpy(smin).cuda()) / float(vsz)).int()
    assert torch.equal(cell, grid[inv])
    cnt = torch.bincount(inv, minlength=V)
    mean = torch.zeros((V, 3), device="cuda", dtype=torch.float64).index_add_(0, inv, xyz.double()) / cnt[:, None]
    assert float((cen.double() - mean).abs().max()) < 1e-5
    # edge cases: empty cloud, one point, unsupported call form
    e = construct_vox_points_closest(torch.zeros((0, 3), device="cuda"), 100)
    assert e[0].shape == (0, 3) and e[2].numel() == 0
    with pytest.raises(HnrError):
        construct_vox_points_closest(xyz[:10], 100, space_min=torch.zeros(3))
    with pytest.raises(HnrError):
        construct_vox_points_closest(xyz[:10].cpu(), 100)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_golden_scatter_outputs_satisfy_the_definition_of_torch_scatter(tag):
    """The image has no torch_scatter, so the golden ran the reference function with stand-ins for its two calls (make_golden.py::gen_voxel).
    This pins those stand-ins -- and with them the stored `centroid` / `min_idx` -- to the DEFINITION of the two operators
    (torch_scatter docs: scatter_mean = per-index arithmetic mean; scatter_min = per-index minimum and an index attaining it), evaluated
    here by brute force in float64 over every voxel: membership from the stored cells, not from any code under test."""
    z = np.load(os.path.join(GOLD, "voxel_down.npz"))
    xyz, grid, cen, midx = z[tag + "_xyz"].astype(np.float64), z[tag + "_grid"], z[tag + "_centroid"], z[tag + "_min_idx"]
    V = grid.shape[0]
    assert cen.shape == (V, 3) and midx.shape == (V,)
    # voxel of every point, re-derived from the stored outputs alone: the chosen point of voxel v lies in v, and a point belongs to the
    # voxel whose cell equals floor((p - space_min) / size); recover (space_min, size) from the data the way :541-551 define them
    x32 = z[tag + "_xyz"]
    mn, mx = x32.min(0), x32.max(0)
    edge = np.float32(np.max(mx - mn) * np.float32(1.05))
    space_min = ((mx + mn) / np.float32(2) - edge / np.float32(2)).astype(np.float32)
    size = np.float32(edge / np.float32(float(z[tag + "_res"][0])))
    cell = np.floor((x32 - space_min[None]) / size).astype(np.int64)
    key = {tuple(c): v for v, c in enumerate(grid.tolist())}
    inv = np.array([key[tuple(c)] for c in cell.tolist()])
    assert len(key) == V and np.array_equal(np.unique(inv), np.arange(V))            # torch.unique(dim=0): every voxel non-empty, no duplicate
    assert all(grid[i].tolist() < grid[i + 1].tolist() for i in range(V - 1))       # ... in lexicographic order
    worst_c, ties = 0.0, 0
    for v in range(V):
        members = np.nonzero(inv == v)[0]
        mean = xyz[members].mean(axis=0)                                             # scatter_mean
        worst_c = max(worst_c, float(np.abs(mean - cen[v]).max()))
        r = np.linalg.norm(xyz[members] - cen[v].astype(np.float64), axis=1)        # residual to the (stored fp32) centroid, :555-556
        assert midx[v] in members                                                    # scatter_min's argument index lies in the group ...
        assert r[list(members).index(midx[v])] <= r.min() * (1 + 1e-6) + 1e-9       # ... and attains the minimum (fp32 rounding of the norm)
        ties += int((r <= r.min() * (1 + 1e-6) + 1e-9).sum() > 1)
    assert worst_c < 2e-6, worst_c
    print("voxel golden %s: %d voxels, centroid max deviation from the float64 mean %.1e, %d voxels with a residual tie" % (tag, V, worst_c, ties))
