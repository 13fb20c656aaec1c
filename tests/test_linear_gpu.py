"""hnr_linear_f32 (fp32 MFMA dense layer) vs an fp64 CPU reference."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K,lda,act", [
    (1000, 256, 284, 284, True), (777, 256, 263, 264, True), (128, 256, 256, 256, True), (1, 256, 284, 284, False),
    (513, 128, 280, 280, True), (300, 128, 128, 128, True), (999, 64, 176, 176, True), (130, 64, 64, 64, False),
    (257, 45, 90, 92, True), (64, 45, 45, 48, False), (5000, 256, 256, 264, True),
])
def test_linear_matches_fp64(M, N, K, lda, act):
    from hybridneuralrendering_amd.linear import PackedLinear
    g = torch.Generator().manual_seed(M * 7 + N + K)
    A = torch.randn(M, lda, generator=g)
    A[:, K:] = float("nan")                     # pad columns must never be read into the product
    W = torch.randn(N, K, generator=g) / np.sqrt(K)
    b = torch.randn(N, generator=g)
    ref = A[:, :K].double() @ W.double().t() + b.double()
    if act:
        ref = torch.nn.functional.leaky_relu(ref, 0.01)
    d = torch.device("cuda:0")
    lin = PackedLinear(W.to(d), b.to(d))
    ldc = N + 3
    out = torch.full((M, ldc), -7.0, device=d)
    lin(A.to(d), out=out, act=act)
    got = out[:, :N].cpu().double()
    err = (got - ref).abs().max().item()
    assert err < 2e-5, err
    assert torch.all(out[:, N:] == -7.0)        # nothing written past N
    # tolerance statement: fp32 fmaf chain, |err| <= ~1e-6 * sum|a*w|


def test_linear_rejects_bad_arguments():
    from hybridneuralrendering_amd.linear import PackedLinear
    from hybridneuralrendering_amd._lib import HnrError
    d = torch.device("cuda:0")
    lin = PackedLinear(torch.randn(64, 90, device=d), None)
    with pytest.raises(HnrError):
        lin(torch.randn(10, 90, device=d))       # lda = 90 is not a multiple of 4
    out = lin(torch.zeros(0, 92, device=d))      # M = 0 is a no-op
    assert out.shape == (0, 64)
