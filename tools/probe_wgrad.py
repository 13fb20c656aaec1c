"""hnr_h2wgrad (256 x 256, the C3 batch's 306 k row slots) and hnr_h2lin alone: for rocprofv3 --pmc passes."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridneuralrendering_amd import _lib  # noqa: E402
from tests.test_h2gemm_gpu import pack  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
M, N, K = 306000, 256, 256
Z, X = torch.randn((M, N), device=dev), torch.randn((M, K), device=dev)
mz = torch.tensor([np.float32(8.0).view(np.int32)], dtype=torch.int32, device=dev)
scratch = torch.empty((int(L.hnr_h2wgrad_scratch_bytes(N, K)),), dtype=torch.uint8, device=dev)
dW, db = torch.empty((N, K), device=dev), torch.empty((N,), device=dev)
img = pack([torch.randn((N, K), device=dev) / 16])[0]
C = torch.empty((M, N), device=dev)
for _ in range(5):
    _lib.check(L.hnr_h2wgrad(_lib.ptr(Z), N, _lib.ptr(X), K, M, None, 1, 0, N, K, _lib.ptr(mz), _lib.ptr(mz), _lib.ptr(dW), K, _lib.ptr(db), 0, _lib.ptr(scratch), _lib.stream()), "wgrad")
    _lib.check(L.hnr_h2lin(_lib.ptr(X), K, M, None, 1, 0, _lib.ptr(img), N, K, 1, 1, 0.01, _lib.ptr(Z), N, _lib.ptr(C), N, None, _lib.stream()), "h2lin")
torch.cuda.synchronize()
