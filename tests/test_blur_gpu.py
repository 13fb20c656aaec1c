"""Blur-handling module (pre-defined kernels) on the GPU vs the reference-generated golden and vs the oracle."""
import os

import numpy as np
import pytest
import torch

from tests.golden_io import GOLD

pytestmark = pytest.mark.gpu


def test_blur_select_matches_reference_golden():
    from hybridneuralrendering_amd.blur import blur_update_output
    z = np.load(os.path.join(GOLD, "blur_select.npz"))
    pn, ps, N, ks = (int(v) for v in z["dims"])
    col = torch.from_numpy(z["color"]).cuda().requires_grad_(True)
    out, sel = blur_update_output(col, torch.from_numpy(z["gt"]).cuda(), torch.from_numpy(z["kernels"]).cuda()[None], pn, ps, return_select=True)
    np.testing.assert_allclose(out.detach().cpu().numpy(), z["out"], rtol=0, atol=2e-6)
    (out * torch.from_numpy(z["upstream"]).cuda()).sum().backward()
    np.testing.assert_allclose(col.grad.cpu().numpy(), z["grad_color"], rtol=0, atol=2e-6)
    assert sel.shape == (pn * pn,) and int(sel.max()) <= N


@pytest.mark.parametrize("pn,ps,N,ks", [(7, 8, 12, 9), (3, 4, 5, 3), (1, 16, 31, 9), (5, 8, 0, 9)])
def test_blur_select_matches_oracle(pn, ps, N, ks):
    from hybridneuralrendering_amd.blur import blur_update_output
    from oracle import render_oracle as ro
    g = torch.Generator().manual_seed(pn * 100 + N)
    S = pn * ps
    col = torch.rand((1, S * S, 3), generator=g)
    gt = torch.rand((1, S * S, 3), generator=g)
    k = torch.rand((max(N, 1), ks, ks), generator=g) ** 4
    k = (k / k.sum(dim=(1, 2), keepdim=True))[:N]
    up = torch.randn((1, S * S, 3), generator=g)
    c0 = col.clone().requires_grad_(True)
    if N > 0:
        ref, rsel = ro.blur_update_output(c0, gt, k, pn, ps)
    else:
        ref, rsel = c0 * 1.0, torch.zeros(pn * pn, dtype=torch.long)
    (ref * up).sum().backward()
    c1 = col.cuda().requires_grad_(True)
    kk = k.cuda()[None] if N > 0 else torch.zeros((1, 0, ks, ks), device="cuda")
    out, sel = blur_update_output(c1, gt.cuda(), kk, pn, ps, return_select=True)
    (out * up.cuda()).sum().backward()
    assert torch.equal(sel.cpu().long(), rsel)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(c1.grad.cpu().numpy(), c0.grad.numpy(), rtol=0, atol=2e-6)


def test_blur_select_patch_major_layout_equals_grid_layout():
    """A rank's whole patches packed (patch, y, x) give the same per-patch results as the grid layout of the full batch."""
    from hybridneuralrendering_amd.blur import blur_update_output
    from hybridneuralrendering_amd.parallel import shard_patches
    z = np.load(os.path.join(GOLD, "blur_select.npz"))
    pn, ps, N, ks = (int(v) for v in z["dims"])
    col, gt, k = torch.from_numpy(z["color"]).cuda(), torch.from_numpy(z["gt"]).cuda(), torch.from_numpy(z["kernels"]).cuda()[None]
    full, sel_full = blur_update_output(col, gt, k, pn, ps, return_select=True)
    for rank in range(3):
        ids, rays = shard_patches(pn, ps, 3, rank)
        rays = rays.cuda()
        c = col[:, rays].clone().requires_grad_(True)
        part, sel = blur_update_output(c, gt[:, rays], k, ids.numel(), ps, return_select=True, layout="patch_major")
        assert torch.equal(part.detach(), full[:, rays]) and torch.equal(sel, sel_full[ids.cuda()])
        part.sum().backward()
        assert torch.isfinite(c.grad).all()


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_learnable_blur_matches_reference_golden(tag):
    """learnable_blur_update_output (HIP grey patches + grouped per-patch convolution with its border rule, forward and backward)
    vs the imported reference method: new colours to 2e-6, gradients w.r.t. the colours and every predictor parameter."""
    from types import SimpleNamespace
    from tests.golden_io import blur_learn_case
    from hybridneuralrendering_amd.blur import learnable_blur_update_output
    cfg, color, gt, up, predictor, blocks, exp = blur_learn_case(tag, device="cuda")
    opt = SimpleNamespace(learnable_blur_kernel_size=cfg["ks"], learnable_blur_kernel_conv=cfg["conv"], learnable_blur_kernel_norm=cfg["norm"],
                          learnable_blur_kernel_mode=cfg["mode"], boundary_mode=cfg["bmode"])
    col = color.clone().requires_grad_(True)
    out = learnable_blur_update_output(col, gt, predictor, opt, cfg["pn"], cfg["ps"])
    (out * up).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), exp["out"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(col.grad.cpu().numpy(), exp["grad_color"], rtol=0, atol=5e-6)
    for bi, blk in enumerate(blocks):
        for k, v in blk.named_parameters():
            g = exp["grads"]["%d.%s" % (bi, k)]
            np.testing.assert_allclose(v.grad.cpu().numpy(), g, rtol=0, atol=2e-5 * max(1.0, float(np.abs(g).max())))


def test_learnable_blur_patch_major_equals_grid_and_checks_arguments():
    from types import SimpleNamespace
    from tests.golden_io import blur_learn_case
    from hybridneuralrendering_amd.blur import learnable_blur_update_output
    from hybridneuralrendering_amd.parallel import shard_patches
    from hybridneuralrendering_amd._lib import HnrError
    cfg, color, gt, up, predictor, blocks, exp = blur_learn_case("a", device="cuda")
    opt = SimpleNamespace(learnable_blur_kernel_size=cfg["ks"], learnable_blur_kernel_conv=0, learnable_blur_kernel_norm=0,
                          learnable_blur_kernel_mode=4, boundary_mode=1)
    full = learnable_blur_update_output(color, gt, predictor, opt, cfg["pn"], cfg["ps"])
    for rank in range(2):
        ids, rays = shard_patches(cfg["pn"], cfg["ps"], 2, rank)
        rays = rays.cuda()
        part = learnable_blur_update_output(color[:, rays], gt[:, rays], predictor, opt, ids.numel(), cfg["ps"], layout="patch_major")
        assert torch.allclose(part, full[:, rays], rtol=0, atol=1e-6)
    opt.boundary_mode = 3
    with pytest.raises(HnrError):
        learnable_blur_update_output(color, gt, predictor, opt, cfg["pn"], cfg["ps"])
    opt.boundary_mode, opt.learnable_blur_kernel_mode = 1, 2
    with pytest.raises(HnrError):
        learnable_blur_update_output(color, gt, predictor, opt, cfg["pn"], cfg["ps"])
