"""hnr_shipped_loss (masked colour MSE + zero-one regulariser, value and gradients on the device) vs the oracle's restatement
of the reference's compute_losses terms (oracle.render_oracle.shipped_loss, itself pinned to the imported reference by
tests/test_render_oracle.py) and vs the stored reference training step."""
import numpy as np
import pytest
import torch

from tests.golden_io import load_train

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R,frac_valid,fw", [(3136, 0.97, None), (3136, 0.4, 0.37), (100, 0.0, None), (200001, 0.9, 1.0)])
def test_shipped_loss_matches_oracle(R, frac_valid, fw):
    from hybridneuralrendering_amd.losses import shipped_loss
    from oracle import render_oracle as ro
    g = torch.Generator().manual_seed(R)
    col = torch.rand((1, R, 3), generator=g)
    gt = torch.rand((1, R, 3), generator=g)
    mask = (torch.rand((1, R), generator=g) < frac_valid).to(torch.int8)
    conf = torch.rand((1, max(int(R * frac_valid), 1), 24, 8), generator=g)
    conf[conf < 0.1] = 0.0                                   # empty slots: clamped to eps, no gradient
    conf[0, 0, 0, 0] = 1.0
    eps = 1e-3
    c0, x0 = col.clone().requires_grad_(True), conf.clone().requires_grad_(True)
    if frac_valid > 0:
        tot, lc, lz = ro.shipped_loss(c0, mask, x0, gt, eps)
    else:                                                    # no valid ray: the reference sets the colour loss to 0 (:1144)
        lc = torch.zeros(())
        lz = torch.mean(torch.log(torch.clamp(x0, eps, 1 - eps)) + torch.log(1 - torch.clamp(x0, eps, 1 - eps)))
        tot = lc + 1e-4 * lz
    scale = 1.0 if fw is None else fw
    ref_total = (lc + 1e-6) * scale + 1e-4 * lz
    ref_total.backward()
    c1, x1 = col.cuda().requires_grad_(True), conf.cuda().requires_grad_(True)
    total, parts = shipped_loss(c1, x1, gt.cuda(), mask.cuda(), eps, 1.0, 1e-4, frame_weight=None if fw is None else torch.tensor([fw]))
    (total * 1.0).backward()
    p = parts.cpu().numpy()
    rt = float(ref_total.detach())
    lc, lz = lc.detach(), lz.detach()
    assert abs(p[0] - rt) <= 2e-6 * max(1.0, abs(rt))
    assert abs(p[1] - float(lc)) <= 2e-6 * max(1e-3, float(lc)) and abs(p[2] - float(lz)) <= 2e-6 * abs(float(lz))
    assert int(p[3]) == int(mask.sum())
    gc = c0.grad if c0.grad is not None else torch.zeros_like(col)
    np.testing.assert_allclose(c1.grad.cpu().numpy(), gc.numpy(), rtol=1e-5, atol=1e-10)
    np.testing.assert_allclose(x1.grad.cpu().numpy(), x0.grad.numpy(), rtol=1e-5, atol=1e-12)


def test_shipped_loss_reproduces_the_reference_training_step_loss():
    from hybridneuralrendering_amd.losses import shipped_loss
    d = load_train("scannet_small")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    total, parts = shipped_loss(t(d["full_coarse_raycolor"]), t(d["conf_coefficient"]), t(d["gt"]), t(d["q_ray_mask"].astype(np.int8)),
                                float(d["zero_epsilon"]))
    ref = d["loss"]                                           # (total, colour, zero-one) of the imported reference's terms
    p = parts.cpu().numpy()
    assert abs(p[1] - ref[1]) <= 2e-6 * ref[1] and abs(p[2] - ref[2]) <= 2e-6 * abs(ref[2])
    assert abs(float(total) - (ref[0] + 1e-6)) <= 2e-6 * abs(ref[0])


@pytest.mark.parametrize("R,frac_valid", [(3136, 0.9), (777, 0.3), (64, 0.0)])
def test_shipped_loss_over_batch_rows_equals_the_masked_copy(R, frac_valid):
    """conf_rows=True (hnr_shipped_loss_rows: conf_coefficient with one row per ray of the batch, rows of invalid rays left out on the
    device) gives the values and gradients of indexing with the mask first."""
    from hybridneuralrendering_amd.losses import shipped_loss, shipped_loss_grads
    g = torch.Generator().manual_seed(R + 5)
    col, gt = torch.rand((R, 3), generator=g).cuda(), torch.rand((R, 3), generator=g).cuda()
    mask = (torch.rand((R,), generator=g) < frac_valid).to(torch.int8).cuda()
    conf = torch.rand((R, 24, 8), generator=g).cuda()
    conf[conf < 0.1] = 0.0
    eps = 1e-3
    c1, x1 = col.clone().requires_grad_(True), conf.clone().requires_grad_(True)
    if frac_valid > 0:
        tot1, p1 = shipped_loss(c1, x1[mask > 0], gt, mask, eps, 1.0, 1e-4)
        tot1.backward()
    c2, x2 = col.clone().requires_grad_(True), conf.clone().requires_grad_(True)
    tot2, p2 = shipped_loss(c2, x2, gt, mask, eps, 1.0, 1e-4, conf_rows=True)
    tot2.backward()
    parts, g_c, g_x = shipped_loss_grads(col, conf, gt, mask, eps, 1.0, 1e-4)
    assert torch.equal(parts, p2) and torch.equal(g_c.reshape(R, 3), c2.grad) and torch.equal(g_x.reshape(conf.shape), x2.grad)
    if frac_valid > 0:
        np.testing.assert_allclose(p1.cpu().numpy(), p2.cpu().numpy(), rtol=1e-6)      # (double partial sums, grouped differently)
        np.testing.assert_allclose(c1.grad.cpu().numpy(), c2.grad.cpu().numpy(), rtol=1e-6, atol=0)
        np.testing.assert_allclose(x1.grad.cpu().numpy(), x2.grad.cpu().numpy(), rtol=1e-6, atol=0)
    else:
        assert float(p2[2]) == 0.0 and float(x2.grad.abs().max()) == 0.0
