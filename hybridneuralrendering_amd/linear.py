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


class FusedMlp3:
    """Three nn.Linear (+LeakyReLU) layers of width <= 128 as ONE launch (hnr_mlp3_forward; csrc/mlp.hip), fp32 in / out.
    A fourth (weight, bias, act) is a TAIL layer on layer 2's output whose result goes to a second output tensor."""

    def __init__(self, weights, biases, acts):
        L = _lib.lib()
        ws = [_lib.require_gpu(w.detach(), "weight", torch.float32) for w in weights]
        bs = [None if b is None else _lib.require_gpu(b.detach(), "bias", torch.float32) for b in biases]
        self.n = len(ws)
        if self.n not in (3, 4):
            raise HnrError("FusedMlp3: 3 layers (+ 1 tail) expected")
        self.N = [int(w.shape[0]) for w in ws]
        self.K = [int(w.shape[1]) for w in ws]
        self.act = [1 if a else 0 for a in acts]
        IN = ctypes.c_int * self.n
        self._N, self._K, self._act = IN(*self.N), IN(*self.K), IN(*self.act)
        nbytes = int(L.hnr_mlp3_packed_bytes(self.n, self._K))
        if nbytes <= 0:
            raise HnrError("FusedMlp3: unsupported layer sizes %r" % (list(zip(self.N, self.K)),))
        dev = ws[0].device
        self.packed = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        PN = ctypes.c_void_p * self.n
        with torch.cuda.device(dev):
            _lib.check(L.hnr_mlp3_pack(self.n, PN(*[w.data_ptr() for w in ws]), IN(*[int(w.stride(0)) for w in ws]), self._N, self._K,
                                       PN(*[(b.data_ptr() if b is not None else None) for b in bs]), _lib.ptr(self.packed), _lib.stream()),
                       "hnr_mlp3_pack")
        self._keep = (ws, bs)

    def __call__(self, a, out, rows_cap, counts=None, count_index=0, count_mult=1, slope=0.01, R=None, ridx=None, seg_stride=0, out2=None):
        L = _lib.lib()
        lda = a.stride(0) if a.shape[0] > 1 else a.shape[1]
        ldc = out.stride(0) if out.shape[0] > 1 else out.shape[1]
        if (self.n == 4) != (out2 is not None):
            raise HnrError("FusedMlp3: a 4-layer stack needs out2 (and only it)")
        ldc2 = 0 if out2 is None else (out2.stride(0) if out2.shape[0] > 1 else out2.shape[1])
        with torch.cuda.device(a.device):
            _lib.check(L.hnr_mlp3_forward(ctypes.c_void_p(a.data_ptr()), int(lda), int(rows_cap), _lib.ptr(counts) if counts is not None else None,
                                          int(count_index), int(count_mult), int(seg_stride), _lib.ptr(self.packed), self.n, self._N, self._K, self._act,
                                          float(slope), _lib.ptr(R) if R is not None else None, _lib.ptr(ridx) if ridx is not None else None,
                                          int(R.stride(0)) if R is not None else 0, ctypes.c_void_p(out.data_ptr()), int(ldc),
                                          ctypes.c_void_p(out2.data_ptr()) if out2 is not None else None, int(ldc2), _lib.stream()),
                       "hnr_mlp3_forward")
        return out
