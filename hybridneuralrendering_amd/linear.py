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

    def side(self, a, R, ridx=None, r_cols=None, r_mode=0, out=None, act=False, slope=0.01, K=None):
        """hnr_linear_f32_side: r_mode 0 -> out = act(a W^T + b + R[ridx[m]]) on the first r_cols columns (R is out itself for "+=");
        r_mode 1 -> out = (a W^T + b) * LeakyReLU'(R[m]) on the first r_cols columns (backward: input gradient)."""
        L = _lib.lib()
        M, lda = a.shape[0], a.stride(0) if a.shape[0] > 1 else a.shape[1]
        if out is None:
            out = torch.empty((M, self.N), dtype=torch.float32, device=a.device)
        ldc = out.stride(0) if M > 1 else out.shape[1]
        ldr = R.stride(0) if R.shape[0] > 1 else R.shape[1]
        with torch.cuda.device(a.device):
            _lib.check(L.hnr_linear_f32_side(ctypes.c_void_p(a.data_ptr()), int(lda), _lib.ptr(self.wp), _lib.ptr(self.bp),
                                             ctypes.c_void_p(R.data_ptr()), _lib.ptr(ridx) if ridx is not None else None, int(ldr),
                                             int(self.N if r_cols is None else r_cols), int(r_mode), ctypes.c_void_p(out.data_ptr()),
                                             int(ldc), M, self.N, self.K if K is None else K, 1 if act else 0, float(slope),
                                             _lib.stream()), "hnr_linear_f32_side")
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


class SplitLinear:
    """A 256-wide nn.Linear on the bf16 matrix cores with exactly split fp32 operands (hnr_linear_s3; csrc/linear_s3.hip).
    Same call surface as PackedLinear.__call__ / gather_add."""

    def __init__(self, weight, bias):
        L = _lib.lib()
        weight = _lib.require_gpu(weight.detach(), "weight", torch.float32)
        self.N, self.K = int(weight.shape[0]), int(weight.shape[1])
        nbytes = int(L.hnr_linear_s3_packed_bytes(self.N, self.K))
        if nbytes <= 0 or self.N != 256:
            raise HnrError("SplitLinear: N must be 256 (got %d)" % self.N)
        dev = weight.device
        self.w3 = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        self.bp = torch.empty((256,), dtype=torch.float32, device=dev)
        b = None if bias is None else _lib.require_gpu(bias.detach(), "bias", torch.float32)
        with torch.cuda.device(dev):
            _lib.check(L.hnr_linear_s3_pack(_lib.ptr(weight), _lib.ptr(b) if b is not None else None, self.N, self.K,
                                            _lib.ptr(self.w3), _lib.ptr(self.bp), _lib.stream()), "hnr_linear_s3_pack")

    def _run(self, a, R, ridx, out, act, slope, K):
        L = _lib.lib()
        if a.dim() != 2 or a.stride(1) != 1:
            raise HnrError("SplitLinear: A must be 2-D with unit inner stride")
        M, lda = a.shape[0], a.stride(0) if a.shape[0] > 1 else a.shape[1]
        if out is None:
            out = torch.empty((M, self.N), dtype=torch.float32, device=a.device)
        ldc = out.stride(0) if M > 1 else out.shape[1]
        with torch.cuda.device(a.device):
            _lib.check(L.hnr_linear_s3(ctypes.c_void_p(a.data_ptr()), int(lda), _lib.ptr(self.w3), _lib.ptr(self.bp),
                                       _lib.ptr(R) if R is not None else None, _lib.ptr(ridx) if ridx is not None else None,
                                       int(R.stride(0)) if R is not None else 0, ctypes.c_void_p(out.data_ptr()), int(ldc), M, self.N,
                                       self.K if K is None else K, 1 if act else 0, float(slope), _lib.stream()), "hnr_linear_s3")
        return out

    def __call__(self, a, out=None, act=False, slope=0.01, K=None):
        return self._run(a, None, None, out, act, slope, K)

    def gather_add(self, a, R, ridx, out=None, act=False, slope=0.01, K=None):
        return self._run(a, R, ridx, out, act, slope, K)


def weight_grad(dZ, X, N, K, dW=None, db=None, accumulate=False, want_bias=True):
    """dW[N,K] = dZ[:, :N]^T X[:, :K], db[N] = column sums of dZ (hnr_linear_f32_wgrad).  dZ / X: 2-D fp32, unit inner
    stride, row strides multiples of 4.  dW may be a column slice of a larger gradient (its row stride is honoured)."""
    L = _lib.lib()
    M = int(dZ.shape[0])
    if X.shape[0] != M:
        raise HnrError("weight_grad: dZ and X must have the same number of rows")
    dev = dZ.device
    if dW is None:
        dW = torch.empty((N, K), dtype=torch.float32, device=dev)
        accumulate = False
    if db is None and want_bias:
        db = torch.empty((N,), dtype=torch.float32, device=dev)
    ldz = dZ.stride(0) if M > 1 else dZ.shape[1]
    ldx = X.stride(0) if M > 1 else X.shape[1]
    scratch = torch.empty((max(int(L.hnr_linear_wgrad_scratch_elems(M, N, K)), 1),), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.hnr_linear_f32_wgrad(ctypes.c_void_p(dZ.data_ptr()), int(ldz), ctypes.c_void_p(X.data_ptr()), int(ldx), M, N, K,
                                          ctypes.c_void_p(dW.data_ptr()), int(dW.stride(0)), _lib.ptr(db) if db is not None else None,
                                          1 if accumulate else 0, _lib.ptr(scratch), _lib.stream()), "hnr_linear_f32_wgrad")
    return dW, db


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
