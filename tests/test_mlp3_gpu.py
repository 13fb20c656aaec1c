"""hnr_mlp3_forward (csrc/mlp.hip): three fused dense layers, fp32 in / out on the two-term fp16 split arithmetic, against an fp64
evaluation of the same nn.Linear + LeakyReLU stack beside the per-layer fp32-MFMA kernel (hnr_linear_f32)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = {
    # name: (K0, N0, N1, N2, acts, with_addend)  -- the three per-sample MLPs of viewmlp (point_aggregators.py:1028-1037, :1199, :1285-1292)
    "color_feature": (280, 128, 128, 128, (1, 1, 1), False),
    "merge_weight": (48, 64, 64, 64, (1, 1, 1), True),
    "mixup": (90, 45, 45, 45, (1, 1, 0), False),
}


def _case(name, M, seed, scale_rows=False):
    from hybridneuralrendering_amd.linear import FusedMlp3, PackedLinear
    K0, N0, N1, N2, acts, add = CASES[name]
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(seed)
    dims = [(N0, K0), (N1, N0), (N2, N1)]
    Ws = [(torch.rand(d, generator=g) * 2 - 1) * (3.0 / d[1]) ** 0.5 for d in dims]
    bs = [(torch.rand(d[0], generator=g) - 0.5) * 0.2 for d in dims]
    if add:
        bs[0] = None
    lda = (K0 + 3) // 4 * 4
    A = torch.randn((M, lda), generator=g)
    A[:, K0:] = 7.0                                     # padding columns must be ignored
    if scale_rows:
        A *= torch.exp2(torch.randint(-12, 13, (M, 1), generator=g).float())
    R = ridx = None
    if add:
        R = torch.randn((max(M // 4, 1), N0), generator=g) * 0.5
        ridx = torch.randint(0, R.shape[0], (M,), generator=g).to(torch.int32)
    Wd, bd = [w.to(dev) for w in Ws], [None if b is None else b.to(dev) for b in bs]
    Ad = A.to(dev)
    f = FusedMlp3(Wd, bd, acts)
    ldc = (N2 + 3) // 4 * 4
    out = torch.full((M, ldc), float("nan"), device=dev)
    counts = torch.tensor([0, 0, M + 5, 0], dtype=torch.int64, device=dev)        # device-side row count larger than the capacity: capacity wins
    f(Ad, out, M, counts, 2, 1, slope=0.01, R=None if R is None else R.to(dev), ridx=None if ridx is None else ridx.to(dev))
    torch.cuda.synchronize()
    # fp64 reference and the per-layer fp32-MFMA path
    lk = lambda x: torch.where(x > 0, x, x * 0.01)
    x64 = A[:, :K0].double()
    x32 = Ad
    for l in range(3):
        y = x64 @ Ws[l].double().T + (bs[l].double() if bs[l] is not None else 0.0)
        if l == 0 and add:
            y = y + R.double()[ridx.long()]
        x64 = lk(y) if acts[l] else y
        pl = PackedLinear(Wd[l], bd[l])
        o32 = torch.zeros((M, (Ws[l].shape[0] + 3) // 4 * 4), device=dev)          # row strides are multiples of 4 floats
        if l == 0 and add:
            x32 = pl.gather_add(x32, R.to(dev), ridx.to(dev), out=o32, act=bool(acts[l]), slope=0.01, K=K0)
        else:
            x32 = pl(x32, out=o32, act=bool(acts[l]), slope=0.01, K=Ws[l].shape[1])
    ref = x64
    den = ref.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    e_fused = float(((out[:, :N2].cpu().double() - ref).abs() / den).max())
    e_f32 = float(((x32[:, :N2].cpu().double() - ref).abs() / den).max())
    return e_fused, e_f32, out


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("M", [1, 130, 70001])
def test_mlp3_is_fp32_class_against_fp64(name, M):
    e_fused, e_f32, out = _case(name, M, seed=M + len(name))
    assert bool(torch.isfinite(out[:, :CASES[name][3]]).all())
    assert e_fused <= 2.5 * e_f32 + 3e-7, (name, M, e_fused, e_f32)


def test_mlp3_rows_of_very_different_magnitude():
    e_fused, e_f32, _ = _case("color_feature", 4099, seed=3, scale_rows=True)
    assert e_fused <= 2.5 * e_f32 + 3e-7, (e_fused, e_f32)


def test_mlp3_device_side_row_count_and_bad_arguments():
    from hybridneuralrendering_amd.linear import FusedMlp3
    from hybridneuralrendering_amd._lib import HnrError
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    Ws = [torch.randn(d, generator=g).to(dev) * 0.1 for d in ((45, 90), (45, 45), (45, 45))]
    f = FusedMlp3(Ws, [None, None, None], (1, 1, 0))
    A = torch.randn((1000, 92), generator=g).to(dev)
    full = torch.zeros((1000, 48), device=dev)
    f(A, full, 1000)
    part = torch.full((1000, 48), -5.0, device=dev)
    counts = torch.tensor([0, 0, 0, 0, 0, 0, 333], dtype=torch.int64, device=dev)
    f(A, part, 1000, counts, 6, 1)                        # only the first 333 rows exist according to the device counter
    assert torch.equal(part[:333], full[:333]) and bool((part[333:] == -5.0).all())
    with pytest.raises(HnrError):
        FusedMlp3([torch.zeros((200, 90), device=dev), Ws[1], Ws[2]], [None] * 3, (1, 1, 0))       # N > 128
    with pytest.raises(HnrError):
        FusedMlp3([torch.zeros((64, 100), device=dev), torch.zeros((64, 64), device=dev), torch.zeros((64, 64), device=dev)], [None] * 3, (1, 1, 1))(
            torch.zeros((4, 100), device=dev), torch.zeros((4, 64), device=dev), 4)                 # no kernel for these k-step counts
