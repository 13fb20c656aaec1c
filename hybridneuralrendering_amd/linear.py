"""Packed fp32 linear layers on libhnr_hip.so's MFMA kernel (hnr_linear_f32)."""
import ctypes

import torch

from . import _lib
from ._lib import HnrError


class PackedLinear:
    """One nn.Linear's weights in the kernel's zero-padded [N_pad, K_pad] layout (packed once)."""

    def __init__(self, weight, bias):
        L = _lib.lib()
        weight = _lib.require_gpu(weight.detach(), "weight", torch.float32)
        self.N, self.K = int(weight.shape[0]), int(weight.shape[1])
        np_, kp_ = ctypes.c_int(), ctypes.c_int()
        _lib.check(L.hnr_linear_packed_dims(self.N, self.K, ctypes.byref(np_), ctypes.byref(kp_)), "hnr_linear_packed_dims")
        self.N_pad, self.K_pad = np_.value, kp_.value
        dev = weight.device
        self.wp = torch.empty((self.N_pad, self.K_pad), dtype=torch.float32, device=dev)
        self.bp = torch.empty((self.N_pad,), dtype=torch.float32, device=dev)
        b = None if bias is None else _lib.require_gpu(bias.detach(), "bias", torch.float32)
        with torch.cuda.device(dev):
            _lib.check(L.hnr_linear_pack(_lib.ptr(weight), _lib.ptr(b) if b is not None else None, self.N, self.K,
                                         _lib.ptr(self.wp), _lib.ptr(self.bp), _lib.stream()), "hnr_linear_pack")

    def gather_add(self, a, R, ridx, out=None, act=False, slope=0.01, K=None):
        """out[m] = act(a[m] W^T + b + R[ridx[m]])  (hnr_linear_f32_gather_add)."""
        L = _lib.lib()
        M, lda = a.shape[0], a.stride(0) if a.shape[0] > 1 else a.shape[1]
        if out is None:
            out = torch.empty((M, self.N), dtype=torch.float32, device=a.device)
        ldc = out.stride(0) if M > 1 else out.shape[1]
        with torch.cuda.device(a.device):
            _lib.check(L.hnr_linear_f32_gather_add(ctypes.c_void_p(a.data_ptr()), int(lda), _lib.ptr(self.wp), _lib.ptr(self.bp),
                                                   _lib.ptr(R), _lib.ptr(ridx), int(R.stride(0)), ctypes.c_void_p(out.data_ptr()),
                                                   int(ldc), M, self.N, self.K if K is None else K, 1 if act else 0, float(slope),
                                                   _lib.stream()), "hnr_linear_f32_gather_add")
        return out

    def __call__(self, a, out=None, act=False, slope=0.01, K=None):
        """a: [M, lda] fp32 (lda % 4 == 0, lda >= K); out: optional [M, ldc] buffer (ldc >= N). Returns out."""
        L = _lib.lib()
        if a.dim() != 2 or a.stride(1) != 1:
            raise HnrError("PackedLinear: A must be 2-D with unit inner stride")
        M, lda = a.shape[0], a.stride(0) if a.shape[0] > 1 else a.shape[1]
        if out is None:
            out = torch.empty((M, self.N), dtype=torch.float32, device=a.device)
        ldc = out.stride(0) if M > 1 else out.shape[1]
        with torch.cuda.device(a.device):
            _lib.check(L.hnr_linear_f32(ctypes.c_void_p(a.data_ptr()), int(lda), _lib.ptr(self.wp), _lib.ptr(self.bp),
                                        ctypes.c_void_p(out.data_ptr()), int(ldc), M, self.N, self.K if K is None else K,
                                        1 if act else 0, float(slope), _lib.stream()), "hnr_linear_f32")
        return out
