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


def test_gather_add_variant_matches_fp64():
    """C[m] = lrelu(A[m] W^T + b + R[ridx[m]]) -- the split form of block1.0 / aux_merge_weight_block.0."""
    from hybridneuralrendering_amd.linear import PackedLinear
    g = torch.Generator().manual_seed(5)
    M, N, K, lda, NR = 3001, 256, 60, 64, 500
    A = torch.randn(M, lda, generator=g); A[:, K:] = float("nan")
    W = torch.randn(N, K, generator=g) / np.sqrt(K)
    b = torch.randn(N, generator=g)
    R = torch.randn(NR, N, generator=g)
    ridx = torch.randint(0, NR, (M,), generator=g, dtype=torch.int32)
    ref = torch.nn.functional.leaky_relu(A[:, :K].double() @ W.double().t() + b.double() + R.double()[ridx.long()], 0.01)
    d = torch.device("cuda:0")
    lin = PackedLinear(W.to(d), b.to(d))
    out = lin.gather_add(A.to(d), R.to(d), ridx.to(d), act=True)
    assert (out.cpu().double() - ref).abs().max().item() < 2e-5
    # N = 64 shape (merge-weight MLP)
    W2 = torch.randn(64, 48, generator=g) / 7.0
    R2 = torch.randn(NR, 64, generator=g)
    A2 = torch.randn(M, 48, generator=g)
    ref2 = torch.nn.functional.leaky_relu(A2.double() @ W2.double().t() + R2.double()[ridx.long()], 0.01)
    out2 = PackedLinear(W2.to(d), None).gather_add(A2.to(d), R2.to(d), ridx.to(d), act=True)
    assert (out2.cpu().double() - ref2).abs().max().item() < 2e-5


def test_point_rows_and_split_equal_unsplit_layer():
    """[emb | PE(emb)] table + gathered addend reproduces the un-split 284-wide first layer of block1."""
    from oracle import render_oracle as ro
    from hybridneuralrendering_amd import scenes, _lib
    from hybridneuralrendering_amd.aggregator import PointAggregator
    d = torch.device("cuda:0")
    torch.manual_seed(3)
    agg = PointAggregator(scenes.default_opt()).to(d)
    emb = (torch.randn(700, 32) * 0.3)
    E_ref = torch.cat([emb, ro.positional_encoding(emb, 3)], dim=-1)            # reference row layout (networks.py:175-189)
    E = torch.empty((700, 224), device=d)
    _lib.check(_lib.lib().hnr_point_rows(_lib.ptr(emb.to(d)), None, 700, 32, _lib.ptr(E), 224, _lib.stream()), "hnr_point_rows")
    np.testing.assert_allclose(E.cpu().numpy(), E_ref.numpy(), rtol=0, atol=3e-7)
    T = agg.point_table(emb.to(d))
    W = agg.block1[0].weight.detach().cpu().double()
    np.testing.assert_allclose(T.cpu().double().numpy(), (E_ref.double() @ W[:, :224].t()).numpy(), rtol=0, atol=2e-5)
