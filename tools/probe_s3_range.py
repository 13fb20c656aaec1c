"""Probe: accuracy of the split-bf16 dense layer beside the fp32-MFMA kernel when the operands' exponents spread widely."""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from hybridneuralrendering_amd.linear import PackedLinear, SplitLinear
dev = torch.device("cuda:0"); torch.manual_seed(1)
for spread in (0, 8, 16, 24):
    M, K = 20000, 256
    A = torch.randn((M, K), device=dev) * torch.exp2((torch.rand((M, K), device=dev) - 0.5) * 2 * spread)
    W = torch.randn((256, K), device=dev) / 16 * torch.exp2((torch.rand((256, K), device=dev) - 0.5) * 2 * min(spread, 8))
    b = torch.zeros(256, device=dev)
    ref = A.double() @ W.double().t()
    mag = A.abs().double() @ W.abs().double().t()
    o3 = SplitLinear(W, b)(A); o1 = PackedLinear(W, b)(A)
    print("exponent spread +-%2d: max err / sum|a||w|  split %.3e  fp32-MFMA %.3e ; rms rel err split %.3e fp32 %.3e" % (
        spread, ((o3.double() - ref).abs() / mag).max().item(), ((o1.double() - ref).abs() / mag).max().item(),
        ((o3.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item(), ((o1.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()))
