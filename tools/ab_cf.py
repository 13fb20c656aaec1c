"""The colour-feature launch (280 -> 128^3 + tail 64) of hnr_mlp3_forward on M rows: weight-stationary kernel (HNR_CF_WS=1) vs mlp3_kernel (default).
The switch is read once per process: run once per setting; the printed SHA-1 of the two outputs must agree (bit-identical kernels).
python tools/ab_cf.py [M]"""
import os, sys, hashlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridneuralrendering_amd.linear import FusedMlp3
dev = torch.device("cuda:0")
M = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3271808
g = torch.Generator(device="cpu").manual_seed(3)
Ws = [torch.randn((128, 280), generator=g) / 16, torch.randn((128, 128), generator=g) / 11, torch.randn((128, 128), generator=g) / 11, torch.randn((64, 128), generator=g) / 11]
bs = [torch.randn((w.shape[0],), generator=g) * 0.1 for w in Ws]
f = FusedMlp3([w.to(dev) for w in Ws], [b.to(dev) for b in bs], (1, 1, 1, 0))
gd = torch.Generator(device=dev).manual_seed(5)
A = torch.randn((M, 280), device=dev, generator=gd) * torch.exp2(torch.randint(-6, 7, (M, 1), device=dev, generator=gd).float())
cnt = torch.zeros((16,), dtype=torch.int64, device=dev); cnt[3] = M - 37          # device-side row count below the capacity
C = torch.full((M, 128), 7.0, device=dev); C2 = torch.full((M, 64), 7.0, device=dev)
run = lambda: f(A, C, M, counts=cnt, count_index=3, out2=C2)
for _ in range(2): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
# fp64 reference on a slice
idx = torch.cat([torch.arange(0, 4096, device=dev), torch.arange(M - 37 - 4096, M - 37, device=dev)])
x = A[idx].double()
for i in range(3):
    x = torch.nn.functional.leaky_relu(x @ Ws[i].to(dev).double().t() + bs[i].to(dev).double(), 0.01)
t = x @ Ws[3].to(dev).double().t() + bs[3].to(dev).double()
err = float((C[idx].double() - x).abs().max() / x.abs().max()), float((C2[idx].double() - t).abs().max() / t.abs().max())
untouched = bool((C[M - 37:] == 7.0).all() and (C2[M - 37:] == 7.0).all())
h = hashlib.sha1(C[:M - 37].cpu().numpy().tobytes() + C2[:M - 37].cpu().numpy().tobytes()).hexdigest()
print("HNR_CF_WS=%s M=%d: %.3f ms per launch; max rel err vs fp64 (CF, tail) %.2e %.2e; rows past the count untouched: %s; sha1 %s" % (os.environ.get("HNR_CF_WS", "0"), M, ms, err[0], err[1], untouched, h))
