"""Backward of the reference-view CNN (aux_block_s1..3, base_rendering_model.py:1047-1063 of the reference: six 3x3 convolutions, strides
2 1 2 1 2 1, LeakyReLU after each) through the C ABI (hnr_image_features_bwd) against torch autograd in float64, at image sizes whose pyramid
levels are not multiples of the kernels' 8 / 16 pixel tiles (ragged edge tiles, odd heights under the stride-2 layers)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CH = [(3, 6, 2), (6, 6, 1), (6, 12, 2), (12, 12, 1), (12, 24, 2), (24, 24, 1)]


def _conv_out(n):
    return (n - 1) // 2 + 1


@pytest.mark.parametrize("V,H,W", [(2, 70, 101), (1, 37, 51), (3, 64, 96), (1, 9, 7)])
def test_image_cnn_backward_matches_autograd(V, H, W):
    from hybridneuralrendering_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(V * 1000 + H)
    slope = 0.01
    img = torch.rand((V, H, W, 3), generator=g)
    ws = [torch.randn((co, ci, 3, 3), generator=g) * (1.5 / (ci * 9) ** 0.5) for ci, co, _ in CH]
    bs = [torch.randn((co,), generator=g) * 0.1 for _, co, _ in CH]
    H1, W1 = _conv_out(H), _conv_out(W)
    H2, W2 = _conv_out(H1), _conv_out(W1)
    H3, W3 = _conv_out(H2), _conv_out(W2)
    shapes = [(V, 6, H1, W1), (V, 6, H1, W1), (V, 12, H2, W2), (V, 12, H2, W2), (V, 24, H3, W3), (V, 24, H3, W3)]      # s1a s1 s2a s2 s3a s3
    up = [torch.randn(s, generator=g) if i % 2 == 1 else torch.zeros(s) for i, s in enumerate(shapes)]                  # upstream d s1 / s2 / s3

    # float64 autograd
    w64 = [w.double().requires_grad_(True) for w in ws]
    b64 = [b.double().requires_grad_(True) for b in bs]
    x = img.double().permute(0, 3, 1, 2)
    acts = []
    for (ci, co, st), w, b in zip(CH, w64, b64):
        x = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x, w, b, stride=st, padding=1), slope)
        acts.append(x)
    sum((a * u.double()).sum() for a, u in zip(acts, up)).backward()

    dev = torch.device("cuda:0")
    imgd = img.to(dev)
    wd = [w.to(dev).contiguous() for w in ws]
    bd = [b.to(dev).contiguous() for b in bs]
    n_scr = int(L.hnr_image_features_scratch_elems(V, H, W))
    assert n_scr == sum(int(np.prod(s)) for s in shapes)
    scratch = torch.zeros((n_scr,), dtype=torch.float32, device=dev)
    fm = torch.empty((V, H, W, 48), dtype=torch.float32, device=dev)
    wp = (ctypes.c_void_p * 6)(*[t.data_ptr() for t in wd])
    bp = (ctypes.c_void_p * 6)(*[t.data_ptr() for t in bd])
    _lib.check(L.hnr_image_features(_lib.ptr(imgd), V, H, W, wp, bp, slope, _lib.ptr(scratch), _lib.ptr(fm), _lib.stream()), "hnr_image_features")
    # the forward scratch holds the six activations in the order above
    off = 0
    for s, a in zip(shapes, acts):
        n = int(np.prod(s))
        np.testing.assert_allclose(scratch[off:off + n].view(s).cpu().numpy(), a.detach().float().numpy(), rtol=0, atol=2e-5)
        off += n
    gpyr = torch.cat([u.reshape(-1) for u in up]).to(dev)
    gw = [torch.zeros_like(t) for t in wd]
    gb = [torch.zeros_like(t) for t in bd]
    gwp = (ctypes.c_void_p * 6)(*[t.data_ptr() for t in gw])
    gbp = (ctypes.c_void_p * 6)(*[t.data_ptr() for t in gb])
    _lib.check(L.hnr_image_features_bwd(_lib.ptr(imgd), V, H, W, wp, slope, _lib.ptr(scratch), _lib.ptr(gpyr), gwp, gbp, _lib.stream()),
               "hnr_image_features_bwd")
    torch.cuda.synchronize()
    for i in range(6):
        for got, ref, name in ((gw[i], w64[i].grad, "weight"), (gb[i], b64[i].grad, "bias")):
            ref = ref.float().numpy()
            scale = max(float(np.abs(ref).max()), 1e-6)
            np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=0, atol=2e-5 * scale, err_msg=f"conv{i} {name}")
