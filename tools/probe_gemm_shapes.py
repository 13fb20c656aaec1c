"""Every dense-layer shape of the C3 training step's backward, each launched ALONE (HIP events, 10 launches): weight gradients (hnr_h2wgrad) and
input gradients (hnr_h2lin mode 1) with the step's row counts -- the table that says which kernel is furthest from its own bounds when nothing
shares the chip with it.    python tools/probe_gemm_shapes.py [rows_per_neighbour_layer] [valid_samples]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridneuralrendering_amd import _lib  # noqa: E402
from tests.test_h2gemm_gpu import pack  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
MR = int(sys.argv[1]) if len(sys.argv) > 1 else 306936
S = int(sys.argv[2]) if len(sys.argv) > 2 else 38367
U = 66000
mz = torch.tensor([np.float32(8.0).view(np.int32)], dtype=torch.int32, device=dev)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def wgrad(M, N, K, ldz=None, ldx=None, nseg=1):
    ldz, ldx = ldz or (N + 3) // 4 * 4, ldx or (K + 3) // 4 * 4
    Z, X = torch.randn((M * nseg, ldz), device=dev), torch.randn((M * nseg, ldx), device=dev)
    scr = torch.empty((int(L.hnr_h2wgrad_scratch_bytes(N, K)),), dtype=torch.uint8, device=dev)
    dW, db = torch.empty((N, K), device=dev), torch.empty((N,), device=dev)
    def f():
        _lib.check(L.hnr_h2wgrad(_lib.ptr(Z), ldz, _lib.ptr(X), ldx, M, None, nseg, M, N, K, _lib.ptr(mz), _lib.ptr(mz), _lib.ptr(dW), K, _lib.ptr(db), 0,
                                 _lib.ptr(scr), _lib.stream()), "wgrad")
    ms = timed(f)
    byts = M * nseg * (ldz + ldx) * 4
    flop = 3 * 2.0 * M * nseg * N * K
    print("wgrad  M %7d x%d  N %3d K %3d : %7.3f ms   operands %6.1f MB = %5.3f ms at 8 TB/s   mfma %5.3f ms at 2.5 PF" % (M, nseg, N, K, ms, byts / 1e6, byts / 8e12 * 1e3, flop / 2.5e15 * 1e3))


def dgrad(M, N, K, nseg=1):
    """dX[M,N] = (dZ[M,K] W) * LeakyReLU'(side): W^T packed as the layer"""
    lda, ldc = (K + 3) // 4 * 4, (N + 3) // 4 * 4
    A, side = torch.randn((M * nseg, lda), device=dev), torch.randn((M * nseg, ldc), device=dev)
    img = pack([torch.randn((K, N), device=dev) / 16], transposed=[True])[0]
    C = torch.empty((M * nseg, ldc), device=dev)
    def f():
        _lib.check(L.hnr_h2lin(_lib.ptr(A), lda, M, None, nseg, M, _lib.ptr(img), N, K, 1, 1, 0.01, _lib.ptr(side), ldc, _lib.ptr(C), ldc, None, _lib.stream()), "h2lin")
    ms = timed(f)
    byts = M * nseg * (lda + 2 * ldc) * 4
    flop = 3 * 2.0 * M * nseg * N * K
    print("dgrad  M %7d x%d  N %3d K %3d : %7.3f ms   operands %6.1f MB = %5.3f ms at 8 TB/s   mfma %5.3f ms at 2.5 PF" % (M, nseg, N, K, ms, byts / 1e6, byts / 8e12 * 1e3, flop / 2.5e15 * 1e3))


print("per-neighbour layers (%d row slots)" % MR)
wgrad(MR, 256, 256)
wgrad(MR, 256, 263, ldx=264)
wgrad(MR, 256, 60, ldx=64)
dgrad(MR, 256, 256)
print("per-point table layer (%d touched points)" % U)
wgrad(U, 256, 224)
dgrad(U, 224, 256)
print("colour feature (%d valid samples)" % S)
wgrad(S, 128, 128); wgrad(S, 128, 280); dgrad(S, 128, 128); dgrad(S, 256, 128)
print("merge-weight MLP (%d samples x 4 views)" % S)
wgrad(S, 64, 64, nseg=4); wgrad(S, 64, 48, nseg=4); wgrad(S, 64, 128); dgrad(S, 64, 64, nseg=4); dgrad(S, 48, 64, nseg=4); dgrad(S, 128, 64)
print("mix-up (%d samples)" % S)
wgrad(S, 45, 45, ldz=48, ldx=48); wgrad(S, 45, 90, ldz=48, ldx=92); dgrad(S, 45, 45); dgrad(S, 90, 45)
