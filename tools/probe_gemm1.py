"""One shape of tools/probe_gemm.py (N = K = 256), for schedule experiments on the 8-wave dense-layer kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hybridneuralrendering_amd.linear import PackedLinear
dev = torch.device("cuda:0")
M = int(float(sys.argv[1])) if len(sys.argv) > 1 else 8_000_000
N, K = 256, 256
A = torch.randn(M, K, device=dev)
W = torch.randn(N, K, device=dev) / K ** 0.5
b = torch.randn(N, device=dev)
lin = PackedLinear(W, b)
out = torch.empty(M, N, device=dev)
for _ in range(2):
    lin(A, out=out, act=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 5
for _ in range(n):
    lin(A, out=out, act=True)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
ref = torch.nn.functional.leaky_relu(A[:4096].double() @ W.double().t() + b.double(), 0.01)
err = (out[:4096].double() - ref).abs().max().item()
print("%s M=%d N=%d K=%d: %.3f ms  %.1f TFLOP/s  err %.1e" % (os.environ.get("TAG", ""), M, N, K, ms, 2.0 * M * N * K / ms / 1e9, err))
