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


@pytest.mark.parametrize("M,K,lda,side", [(128, 256, 256, False), (1000, 256, 256, False), (777, 263, 264, False), (5000, 60, 64, True),
                                          (70001, 256, 288, False), (1, 256, 256, False), (129, 40, 64, True)])
def test_split_bf16_layer_is_fp32_class(M, K, lda, side):
    """hnr_linear_s3 (fp32 operands split exactly into three bf16 terms, six bf16 MFMAs, fp32 accumulate) against fp64, beside
    the fp32-MFMA kernel on the same inputs.  Tolerance: the error of an fp32 dot product, |err| <= 4e-7 * (sum|a w| + |b| + |r|)
    (measured 1.5e-7 for both kernels); the two kernels differ from each other by summation order only."""
    from hybridneuralrendering_amd.linear import PackedLinear, SplitLinear
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M + 3 * K)
    A = torch.randn(M, lda, generator=g) * (torch.rand(M, 1, generator=g) * 4)
    W = torch.randn(256, K, generator=g) / np.sqrt(K)
    b = torch.randn(256, generator=g)
    ref = A[:, :K].double() @ W.double().t() + b.double()
    mag = A[:, :K].abs().double() @ W.abs().double().t() + b.abs().double()
    R = ridx = None
    if side:
        R = torch.randn(300, 256, generator=g)
        ridx = torch.randint(0, 300, (M,), generator=g, dtype=torch.int32)
        ref = ref + R.double()[ridx.long()]
        mag = mag + R.abs().double()[ridx.long()]
    ref = torch.nn.functional.leaky_relu(ref, 0.01)
    A2 = A.clone()
    A2[:, K:] = float("nan")                    # pad columns must never enter a product
    s3, f32 = SplitLinear(W.to(d), b.to(d)), PackedLinear(W.to(d), b.to(d))
    ldc = 260
    o3 = torch.full((M, ldc), -7.0, device=d)
    if side:
        s3.gather_add(A2.to(d), R.to(d), ridx.to(d), out=o3, act=True, K=K)
        o1 = f32.gather_add(A2.to(d), R.to(d), ridx.to(d), act=True, K=K)
    else:
        s3(A2.to(d), out=o3, act=True, K=K)
        o1 = f32(A2.to(d), act=True, K=K)
    e3 = ((o3[:, :256].cpu().double() - ref).abs() / mag).max().item()
    e1 = ((o1.cpu().double() - ref).abs() / mag).max().item()
    assert e3 < 4e-7, (e3, e1)
    assert e3 < 2.0 * e1 + 5e-8, (e3, e1)
    assert torch.all(o3[:, 256:] == -7.0)


def test_split_bf16_rejects_bad_arguments():
    from hybridneuralrendering_amd.linear import SplitLinear
    from hybridneuralrendering_amd._lib import HnrError
    d = torch.device("cuda:0")
    with pytest.raises(HnrError):
        SplitLinear(torch.randn(128, 64, device=d), None)          # N must be 256
    lin = SplitLinear(torch.randn(256, 64, device=d), None)
    with pytest.raises(HnrError):
        lin(torch.randn(10, 63, device=d))                          # lda not a multiple of 4
    with pytest.raises(HnrError):
        lin(torch.randn(10, 32, device=d))                          # lda < K


@pytest.mark.parametrize("spread", [8, 24])
def test_split_bf16_layer_keeps_fp32_accuracy_over_a_wide_dynamic_range(spread):
    """Operands whose exponents spread over 2^(-spread) .. 2^(+spread): the three-term split is exact at every scale, so the
    error stays that of the fp32-MFMA kernel (both ~1e-6 * sum|a w| at worst)."""
    from hybridneuralrendering_amd.linear import PackedLinear, SplitLinear
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(spread)
    M, K = 20000, 256
    A = torch.randn((M, K), generator=g) * torch.exp2((torch.rand((M, K), generator=g) - 0.5) * 2 * spread)
    W = torch.randn((256, K), generator=g) / 16 * torch.exp2((torch.rand((256, K), generator=g) - 0.5) * 16)
    b = torch.zeros(256)
    ref = A.double() @ W.double().t()
    mag = A.abs().double() @ W.abs().double().t()
    e3 = ((SplitLinear(W.to(d), b.to(d))(A.to(d)).cpu().double() - ref).abs() / mag).max().item()
    e1 = ((PackedLinear(W.to(d), b.to(d))(A.to(d)).cpu().double() - ref).abs() / mag).max().item()
    assert e3 < 2e-6 and e3 < 1.5 * e1 + 1e-7, (e3, e1)
