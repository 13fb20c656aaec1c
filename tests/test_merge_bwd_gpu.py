"""hnr_merge_bwd (transpose of the merge over the reference views: models/aggregators/point_aggregators.py:1320-1345 in the reference --
sigmoid merge weight from the last 64 -> 1 layer, view mask, optional frame weights, weighted mean of the 45 image-feature columns) against
float64 autograd, for view counts on both of the kernel's compiled forms (V <= 4: 16-wave workgroups; V > 4: the generic form)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("V,S,use_fw", [(4, 700, False), (3, 129, True), (6, 300, True), (1, 65, False)])
def test_merge_backward_matches_autograd(V, S, use_fw):
    from hybridneuralrendering_amd import _lib
    L = _lib.lib()
    cap = S + 13                                     # rows of view v start at v * cap
    SR = 24
    slope = 0.01
    g = torch.Generator().manual_seed(V * 1000 + S)
    X6 = torch.randn(V * cap, 48, generator=g)
    Hm = torch.randn(V * cap, 64, generator=g)
    w_last = torch.randn(64, generator=g) * 0.3
    b_last = torch.randn(1, generator=g) * 0.1
    vmask = (torch.rand(V * cap, generator=g) > 0.25).float()
    frame_w = (torch.rand(V, generator=g) + 0.5) if use_fw else None
    gX7 = torch.randn(S, 92, generator=g)
    gCF0 = torch.randn(S, 128, generator=g)
    vs_item = torch.arange(S, dtype=torch.int32) * 1               # sample s sits at ray s // SR
    counts = torch.zeros(16, dtype=torch.int64)
    counts_idx_valid = 6                                           # HNR_CNT_SAMPLES_VALID (include/hnr.h)
    counts[counts_idx_valid] = S

    rows = (torch.arange(V)[:, None] * cap + torch.arange(S)[None, :])           # [V,S]
    f = X6[rows][..., :45].double().requires_grad_(True)
    h = Hm[rows].double().requires_grad_(True)
    w64 = w_last.double().requires_grad_(True)
    b64 = b_last.double().requires_grad_(True)
    sg = torch.sigmoid((h * w64).sum(-1) + b64)
    scale = vmask[rows].double() * (frame_w.double()[:, None] if use_fw else 1.0)
    wv = sg * scale
    merged = (f * wv[..., None]).sum(0) / (wv.sum(0) + 1e-6)[:, None]
    (merged * gX7[:, 45:90].double()).sum().backward()
    ref_gF = f.grad.float().numpy()
    ref_gZ3 = (h.grad * torch.where(h.detach() > 0, 1.0, slope)).float().numpy()

    dev = torch.device("cuda:0")
    t = lambda x: x.to(dev).contiguous()
    dX6, dHm, dw, db, dvm, dgX7, dvs, dcnt = t(X6), t(Hm), t(w_last), t(b_last), t(vmask), t(gX7), t(vs_item), t(counts)
    dfw = t(frame_w) if use_fw else None
    gF = torch.full((V * cap, 48), 3.0, device=dev)
    gZ3 = torch.full((V * cap, 64), 3.0, device=dev)
    gCF = t(gCF0)
    gw = torch.zeros(64, device=dev)
    gb = torch.zeros(1, device=dev)
    _lib.check(L.hnr_merge_bwd(_lib.ptr(dX6), 48, _lib.ptr(dHm), 64, _lib.ptr(dw), _lib.ptr(db), _lib.ptr(dvm), _lib.ptr(dfw) if use_fw else None, _lib.ptr(dcnt), V, cap, slope,
                               None, _lib.ptr(dvs), SR, _lib.ptr(dgX7), 92, _lib.ptr(gF), 48, _lib.ptr(gZ3), 64, _lib.ptr(gCF), 128, _lib.ptr(gw), _lib.ptr(gb), _lib.stream()),
               "hnr_merge_bwd")
    got_gF = gF.cpu()[rows][..., :45].numpy()
    got_gZ3 = gZ3.cpu()[rows].numpy()
    tol = lambda r: 3e-5 * max(1.0, float(np.abs(r).max()))
    np.testing.assert_allclose(got_gF, ref_gF, rtol=2e-4, atol=tol(ref_gF))
    np.testing.assert_allclose(got_gZ3, ref_gZ3, rtol=2e-4, atol=tol(ref_gZ3))
    np.testing.assert_allclose(gw.cpu().numpy(), w64.grad.float().numpy(), rtol=2e-4, atol=tol(w64.grad.numpy()))
    np.testing.assert_allclose(gb.cpu().numpy(), b64.grad.float().numpy(), rtol=2e-4, atol=tol(w64.grad.numpy()))
    np.testing.assert_allclose(gCF.cpu().numpy()[:, :45], (gCF0[:, :45] + gX7[:, :45]).numpy(), rtol=0, atol=1e-6)          # d colfeat[:45] is ADDED
    np.testing.assert_array_equal(gCF.cpu().numpy()[:, 45:], gCF0[:, 45:].numpy())
    assert float(gF.cpu()[rows][..., 45:].abs().max()) == 0.0                                                                # padding columns 45..47 = 0
