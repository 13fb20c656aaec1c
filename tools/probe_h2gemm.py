"""Timing of the training-step GEMM kernels (csrc/h2gemm.hip) at the C3 batch's sizes: 306 k padded neighbour rows, 38 k samples."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridneuralrendering_amd import _lib  # noqa: E402
from tests.test_h2gemm_gpu import pack  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")


def timed(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for M, N, K in [(306000, 256, 256), (306000, 256, 264), (306000, 256, 64), (100000, 256, 224), (38000, 128, 280), (38000, 128, 128), (153000, 64, 64), (153000, 64, 48), (38000, 45, 90)]:
    ldz, ldx = (N + 3) // 4 * 4, (K + 3) // 4 * 4
    Z, X = torch.randn((M, ldz), device=dev), torch.randn((M, ldx), device=dev)
    mz = torch.tensor([np.float32(8.0).view(np.int32)], dtype=torch.int32, device=dev)
    scratch = torch.empty((int(L.hnr_h2wgrad_scratch_bytes(N, min(K, 287))),), dtype=torch.uint8, device=dev)
    dW, db = torch.empty((N, K), device=dev), torch.empty((N,), device=dev)
    Kw = min(K, 287)
    ms = timed(lambda: _lib.check(L.hnr_h2wgrad(_lib.ptr(Z), ldz, _lib.ptr(X), ldx, M, None, 1, 0, N, Kw, _lib.ptr(mz), _lib.ptr(mz), _lib.ptr(dW), K, _lib.ptr(db), 0,
                                               _lib.ptr(scratch), _lib.stream()), "wgrad"))
    print("wgrad  M=%6d N=%3d K=%3d  %.3f ms  %.1f TFLOP/s(fp32-equivalent)  %.2f TB/s" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9, M * (ldz + ldx) * 4 / ms / 1e9))
    if K in (256, 224, 128, 64, 48) or K <= 48:
        W = torch.randn((N, K), device=dev) / 16
        img = pack([W])[0]
        C = torch.empty((M, ldz), device=dev)
        side = torch.randn((M, ldz), device=dev)
        for mode in (0, 1):
            ms = timed(lambda: _lib.check(L.hnr_h2lin(_lib.ptr(X), ldx, M, None, 1, 0, _lib.ptr(img), N, K, mode, 1, 0.01, _lib.ptr(side), ldz, _lib.ptr(C), ldz, None,
                                                     _lib.stream()), "h2lin"))
            print("h2lin%d M=%6d N=%3d K=%3d  %.3f ms  %.1f TFLOP/s(fp32-equivalent)  %.2f TB/s" % (mode, M, N, K, ms, 2.0 * M * N * K / ms / 1e9, M * (ldz * (1 + mode) + ldx) * 4 / ms / 1e9))
