"""Does hnr_h2lin give the same bits when an hnr_h2wgrad of a small layer runs beside it on another stream?"""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridneuralrendering_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_h2gemm_gpu import pack
L = _lib.lib(); dev = torch.device("cuda:0"); p = _lib.ptr
M = 38391
g = torch.Generator().manual_seed(1)
A = torch.randn((M, 128), generator=g).to(dev); W = (torch.randn((256, 128), generator=g) / 11).to(dev)
img = pack([W])[0]
C0 = torch.empty((M, 256), device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def lin(out, st):
    _lib.check(L.hnr_h2lin(p(A), 128, M, None, 1, 0, p(img), 256, 128, 0, 0, 0.01, None, 0, p(out), 256, None, ctypes.c_void_p(st.cuda_stream)), "lin")
torch.cuda.synchronize()
lin(C0, sA); torch.cuda.synchronize()
for (N, K) in ((128, 128), (45, 45), (64, 64), (128, 280), (256, 256)):
    Z = torch.randn((M, (N + 3) // 4 * 4), generator=g).to(dev); X = torch.randn((M, (K + 3) // 4 * 4), generator=g).to(dev)
    one = torch.tensor([np.float32(8.0).view(np.int32)], dtype=torch.int32, device=dev)
    sc = torch.empty((int(L.hnr_h2wgrad_scratch_bytes(N, K)),), dtype=torch.uint8, device=dev)
    dW, db = torch.empty((N, K), device=dev), torch.empty((N,), device=dev)
    dW0, db0 = torch.empty((N, K), device=dev), torch.empty((N,), device=dev)
    def wg(o1, o2, st):
        _lib.check(L.hnr_h2wgrad(p(Z), Z.shape[1], p(X), X.shape[1], M, None, 1, 0, N, K, p(one), p(one), p(o1), K, p(o2), 0, p(sc), ctypes.c_void_p(st.cuda_stream)), "wg")
    wg(dW0, db0, sB); torch.cuda.synchronize()
    bad_l = bad_w = 0
    for it in range(30):
        C = torch.empty((M, 256), device=dev)
        torch.cuda.synchronize()
        wg(dW, db, sB); lin(C, sA); wg(dW, db, sB); lin(C, sA)
        torch.cuda.synchronize()
        bad_l += int(not torch.equal(C, C0)); bad_w += int(not (torch.equal(dW, dW0) and torch.equal(db, db0)))
    print("wgrad N=%d K=%d beside h2lin<8>: h2lin differs %d/30, wgrad differs %d/30" % (N, K, bad_l, bad_w))
