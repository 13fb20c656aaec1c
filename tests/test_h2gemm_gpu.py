"""Dense layers of the training step on the 16-bit matrix pipe (csrc/h2gemm.hip) against fp64 numpy, through the C ABI.

What is checked is what torch autograd computes for an nn.Linear (+ LeakyReLU) of PointAggregator.viewmlp in the reference's train step
(/root/reference/models/aggregators/point_aggregators.py:948, :972, :1037, :1199, :1292): forward, input gradient through the activation,
weight / bias gradient.  Tolerances are those of an fp32 GEMM: |err| <= tol * sum_k |a_k w_k| per element (fp32 MFMA measures ~1.5e-7 here,
tests/test_linear_gpu.py)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hybridneuralrendering_amd import _lib  # noqa: E402


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def pack(W_list, transposed=None, biases=None):
    """hnr_h2lin_pack over a list of torch weights [N,K]; transposed[j]: pack W^T (the input-gradient layer)."""
    L = _lib.lib()
    n = len(W_list)
    transposed = transposed or [False] * n
    biases = biases or [None] * n
    outs, Ns, Ks, rs, cs = [], [], [], [], []
    for W, t in zip(W_list, transposed):
        rows, cols = W.shape
        N, K = (cols, rows) if t else (rows, cols)
        Ns.append(N); Ks.append(K)
        rs.append(1 if t else cols); cs.append(cols if t else 1)
        outs.append(torch.empty((int(L.hnr_h2lin_packed_bytes(K)),), dtype=torch.uint8, device=W.device))
    P, I, I64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    _lib.check(L.hnr_h2lin_pack(n, (P * n)(*[w.data_ptr() for w in W_list]), (I64 * n)(*rs), (I64 * n)(*cs), (I * n)(*Ns), (I * n)(*Ks),
                                (P * n)(*[(b.data_ptr() if b is not None else None) for b in biases]), (P * n)(*[o.data_ptr() for o in outs]),
                                _lib.stream()), "hnr_h2lin_pack")
    return outs


@pytest.mark.parametrize("M,N,K,mode", [(1000, 256, 256, 0), (777, 256, 224, 0), (5000, 256, 256, 1), (300, 224, 256, 1), (2049, 128, 128, 1),
                                        (513, 64, 64, 1), (64, 48, 64, 1), (1500, 128, 45, 0), (4097, 90, 45, 1), (1, 256, 256, 0)])
def test_h2lin_matches_fp64(M, N, K, mode):
    dev = _dev()
    L = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N + K)
    lda, ldc = (K + 3) // 4 * 4, (N + 3) // 4 * 4
    A = torch.zeros((M, lda))
    A[:, :K] = torch.randn((M, K), generator=g) * torch.exp(2.0 * torch.randn((M, 1), generator=g))      # rows of very different magnitude
    W = torch.randn((N, K), generator=g) / np.sqrt(K)
    b = torch.randn((N,), generator=g)
    side = torch.randn((M, ldc), generator=g)
    Ad, Wd, bd, sd = A.to(dev), W.to(dev), b.to(dev), side.to(dev)
    img = pack([Wd], biases=[bd])[0]
    C = torch.full((M + 3, ldc), 7.0, device=dev)
    mx = torch.zeros(1, dtype=torch.int32, device=dev)
    dm = torch.tensor([M], dtype=torch.int64, device=dev)
    slope = 0.01
    _lib.check(L.hnr_h2lin(_lib.ptr(Ad), lda, M + 3, _lib.ptr(dm), 1, 0, _lib.ptr(img), N, K, mode, 1, slope, _lib.ptr(sd) if mode == 1 else None, ldc,
                           _lib.ptr(C), ldc, _lib.ptr(mx), _lib.stream()), "hnr_h2lin")
    got = C.cpu().numpy().astype(np.float64)
    A64, W64 = A[:, :K].numpy().astype(np.float64), W.numpy().astype(np.float64)
    ref = A64 @ W64.T
    mag = np.abs(A64) @ np.abs(W64).T
    if mode == 0:
        ref = ref + b.numpy().astype(np.float64)
        mag = mag + np.abs(b.numpy().astype(np.float64))
        ref = np.where(ref > 0, ref, slope * ref)
    else:
        ref = ref * np.where(side[:, :N].numpy() > 0, 1.0, slope)
    err = np.abs(got[:M, :N] - ref) / (mag + 1e-30)
    assert err.max() < 4e-7, (err.max(), np.unravel_index(err.argmax(), err.shape))
    assert (got[M:] == 7.0).all(), "rows past *d_m were written"
    if mode == 1:
        gmax = float(np.frombuffer(np.int32(mx.item()).tobytes(), dtype=np.float32)[0])
        assert gmax >= np.abs(got[:M, :N]).max() * (1 - 1e-6) and gmax <= np.abs(got[:M, :N]).max() * 1.0001 + 1e-30


def _sign_words(side, M_cap):
    """The chain kernels' sign words of an activation [M_cap, 256] (ChainArgs::hbits): word ((tile * 4 + wave) * 64 + lane), bit 31 - i = (value i of the
    lane's 32 columns 64 wave + 16 (lane >> 5) + {0..15, 32..47} of row 32 tile + (lane & 31) is > 0)."""
    pos = (side[:, :256] > 0).reshape(M_cap // 32, 32, 4, 2, 2, 16)           # tile, row j, wave, column tile c, lane half h, r
    bits = np.zeros((M_cap // 32, 4, 2, 32), dtype=np.uint32)                  # tile, wave, h, j
    for c in range(2):
        for r in range(16):
            bits |= np.transpose(pos[:, :, :, c, :, r], (0, 2, 3, 1)).astype(np.uint32) << np.uint32(31 - (16 * c + r))
    return bits.reshape(-1).view(np.int32)


@pytest.mark.parametrize("M_cap,M,K,lda,ldc", [(4096, 4001, 256, 256, 256), (8192, 8192, 256, 264, 264), (64, 33, 256, 256, 264), (32, 1, 256, 256, 256),
                                               (2048, 2000, 224, 224, 256), (96, 96, 256, 256, 256), (64, 0, 256, 256, 256)])
def test_h2lin_dgrad_bits_matches_fp64_and_the_float_side_form(M_cap, M, K, lda, ldc):
    """hnr_h2lin_dgrad_bits (K = 256: the weight-stationary kernel csrc/h2lin_ws.hip; other K: the streaming kernel reading sign words) = what torch autograd
    computes for dX of a Linear behind a LeakyReLU whose output is known by its signs; equal BIT FOR BIT to hnr_h2lin mode 1 with the activation itself."""
    dev = _dev()
    L = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(M_cap + M + K)
    N = 256
    A = torch.zeros((M_cap, lda))
    A[:, :K] = torch.randn((M_cap, K), generator=g) * torch.exp(2.0 * torch.randn((M_cap, 1), generator=g))
    W = torch.randn((K, N), generator=g) / np.sqrt(K)                          # the forward layer's weight [K outputs of the forward = our K inputs, N]
    side = torch.randn((M_cap, ldc), generator=g)
    Ad, sd = A.to(dev), side.to(dev)
    img = pack([W.to(dev)], transposed=[True])[0]
    bits = torch.from_numpy(_sign_words(side.numpy(), M_cap)).to(dev)
    dm = torch.tensor([M], dtype=torch.int64, device=dev)
    slope = 0.01
    C1, C2 = torch.full((M_cap, ldc), 7.0, device=dev), torch.full((M_cap, ldc), 7.0, device=dev)
    m1, m2 = torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    for _ in range(2):                                                         # (twice: the maximum word is a running maximum, the rows are rewritten)
        _lib.check(L.hnr_h2lin_dgrad_bits(_lib.ptr(Ad), lda, M_cap, _lib.ptr(dm), _lib.ptr(img), N, K, slope, _lib.ptr(bits), _lib.ptr(C1), ldc, _lib.ptr(m1),
                                          _lib.stream()), "hnr_h2lin_dgrad_bits")
    _lib.check(L.hnr_h2lin(_lib.ptr(Ad), lda, M_cap, _lib.ptr(dm), 1, 0, _lib.ptr(img), N, K, 1, 0, slope, _lib.ptr(sd), ldc, _lib.ptr(C2), ldc, _lib.ptr(m2),
                           _lib.stream()), "hnr_h2lin")
    got = C1.cpu().numpy()
    assert got.tobytes() == C2.cpu().numpy().tobytes() and m1.item() == m2.item()
    assert (got[M:] == 7.0).all() and (got[:, N:] == 7.0).all(), "rows past *d_m / padding columns were written"
    if M == 0:                                                                 # no rows: nothing written, the maximum word untouched
        assert m1.item() == 0
        return
    A64, W64 = A[:M, :K].numpy().astype(np.float64), W.numpy().astype(np.float64)
    ref = (A64 @ W64) * np.where(side[:M, :N].numpy() > 0, 1.0, slope)
    mag = np.abs(A64) @ np.abs(W64)
    err = np.abs(got[:M, :N].astype(np.float64) - ref) / (mag + 1e-30)
    assert err.max() < 4e-7, err.max()
    gmax = float(np.frombuffer(np.int32(m1.item()).tobytes(), dtype=np.float32)[0])
    assert gmax == np.abs(got[:M, :N]).max()


@pytest.mark.parametrize("M,N,K", [(20000, 256, 256), (777, 256, 263), (3001, 256, 60), (999, 256, 224), (4096, 128, 280), (5000, 128, 128),
                                   (3333, 64, 48), (2500, 64, 64), (1200, 64, 128), (900, 45, 90), (31, 45, 45), (1, 256, 256)])
def test_h2wgrad_matches_fp64(M, N, K):
    dev = _dev()
    L = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(M + 3 * N + 5 * K)
    ldz, ldx = (N + 3) // 4 * 4, (K + 3) // 4 * 4
    Z = torch.full((M + 5, ldz), float("nan"))
    X = torch.full((M + 5, ldx), float("nan"))
    Z[:M, :N] = torch.randn((M, N), generator=g) * torch.exp(1.5 * torch.randn((M, 1), generator=g)) * 1e-3
    X[:M, :K] = torch.randn((M, K), generator=g).abs() * torch.exp(torch.randn((1, K), generator=g))
    Z[:M, N:] = 3.0; X[:M, K:] = -2.0                                              # padding columns must not leak in
    Zd, Xd = Z.to(dev), X.to(dev)
    dm = torch.tensor([M], dtype=torch.int64, device=dev)
    mz = torch.zeros(1, dtype=torch.int32, device=dev)
    mx = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(L.hnr_absmax(_lib.ptr(Zd), ldz, M + 5, _lib.ptr(dm), 1, 0, N, _lib.ptr(mz), _lib.stream()), "hnr_absmax")
    _lib.check(L.hnr_absmax(_lib.ptr(Xd), ldx, M + 5, _lib.ptr(dm), 1, 0, K, _lib.ptr(mx), _lib.stream()), "hnr_absmax")
    zmax = float(np.frombuffer(np.int32(mz.item()).tobytes(), dtype=np.float32)[0])
    assert zmax == float(Z[:M, :N].abs().max())
    scratch = torch.empty((int(L.hnr_h2wgrad_scratch_bytes(N, K)),), dtype=torch.uint8, device=dev)
    dW = torch.full((N, K + 2), 5.0, device=dev)
    db = torch.full((N,), 5.0, device=dev)
    for acc in (0, 1):
        _lib.check(L.hnr_h2wgrad(_lib.ptr(Zd), ldz, _lib.ptr(Xd), ldx, M + 5, _lib.ptr(dm), 1, 0, N, K, _lib.ptr(mz), _lib.ptr(mx), _lib.ptr(dW), K + 2,
                                 _lib.ptr(db), acc, _lib.ptr(scratch), _lib.stream()), "hnr_h2wgrad")
    Z64, X64 = Z[:M, :N].numpy().astype(np.float64), X[:M, :K].numpy().astype(np.float64)
    ref, mag = Z64.T @ X64, np.abs(Z64).T @ np.abs(X64)
    got = dW.cpu().numpy().astype(np.float64)
    err = np.abs(got[:, :K] / 2.0 - ref) / (mag + 1e-30)
    # (a single product carries the split error of both operands, 2 x 2^-22; over a sum the terms' errors average out)
    assert err.max() < (1e-6 if M < 8 else 4e-7), (err.max(), np.unravel_index(err.argmax(), err.shape))
    assert (got[:, K:] == 5.0).all()
    errb = np.abs(db.cpu().numpy().astype(np.float64) / 2.0 - Z64.sum(0)) / (np.abs(Z64).sum(0) + 1e-30)
    assert errb.max() < 4e-7, errb.max()


def test_h2wgrad_is_bit_identical_run_to_run_and_zero_rows_are_inert():
    dev = _dev()
    L = _lib.lib()
    M, N, K = 30000, 256, 256
    g = torch.Generator(device="cpu").manual_seed(3)
    Z, X = torch.randn((M, N), generator=g), torch.randn((M, K), generator=g)
    Z[::3] = 0.0                                                                   # empty neighbour slots of the padded row layout: exact zeros
    X[::3] = 50.0                                                                  # ... whatever (finite) activations those rows carry
    Zd, Xd = Z.to(dev), X.to(dev)
    one = torch.tensor([np.float32(8.0).view(np.int32)], dtype=torch.int32, device=dev)      # upper bounds of the maxima are enough
    big = torch.tensor([np.float32(64.0).view(np.int32)], dtype=torch.int32, device=dev)
    scratch = torch.empty((int(L.hnr_h2wgrad_scratch_bytes(N, K)),), dtype=torch.uint8, device=dev)
    outs = []
    for _ in range(2):
        dW, db = torch.empty((N, K), device=dev), torch.empty((N,), device=dev)
        _lib.check(L.hnr_h2wgrad(_lib.ptr(Zd), N, _lib.ptr(Xd), K, M, None, 1, 0, N, K, _lib.ptr(one), _lib.ptr(big), _lib.ptr(dW), K, _lib.ptr(db), 0,
                                 _lib.ptr(scratch), _lib.stream()), "hnr_h2wgrad")
        outs.append((dW.cpu().numpy(), db.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    ref = Z.numpy().astype(np.float64).T @ X.numpy().astype(np.float64)
    mag = np.abs(Z.numpy().astype(np.float64)).T @ np.abs(X.numpy().astype(np.float64))
    assert (np.abs(outs[0][0] - ref) / mag).max() < 4e-7


def test_segmented_rows_equal_the_packed_layout():
    """n_seg > 1 (the (view, sample) layout of the merge-weight MLP's rows: segment v starts at physical row v * seg_stride, every segment holds
    *d_m rows): hnr_h2lin, hnr_h2wgrad and hnr_absmax give what they give on the same rows packed back to back."""
    dev = _dev()
    L = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(11)
    n, V, stride, N, K = 1234, 4, 2000, 64, 48
    packed = torch.randn((V * n, K), generator=g) * torch.exp(torch.randn((V * n, 1), generator=g))
    Zp = torch.randn((V * n, N), generator=g) * 1e-2
    seg = torch.full((V * stride, K), float("nan")); segz = torch.full((V * stride, N), float("nan"))
    for v in range(V):
        seg[v * stride:v * stride + n] = packed[v * n:(v + 1) * n]
        segz[v * stride:v * stride + n] = Zp[v * n:(v + 1) * n]
    W = torch.randn((N, K), generator=g) / np.sqrt(K); b = torch.randn((N,), generator=g)
    Wd, bd = W.to(dev), b.to(dev)
    img = pack([Wd], biases=[bd])[0]
    dn = torch.tensor([n], dtype=torch.int64, device=dev)
    dall = torch.tensor([V * n], dtype=torch.int64, device=dev)
    outs = []
    for A, nseg, ss, dm, rows in ((seg.to(dev), V, stride, dn, V * stride), (packed.to(dev), 1, 0, dall, V * n)):
        C = torch.full((rows, N), 7.0, device=dev)
        mx = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(L.hnr_h2lin(_lib.ptr(A), K, stride if nseg > 1 else rows, _lib.ptr(dm), nseg, ss, _lib.ptr(img), N, K, 0, 1, 0.01, None, 0, _lib.ptr(C), N, _lib.ptr(mx),
                               _lib.stream()), "hnr_h2lin")
        am = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(L.hnr_absmax(_lib.ptr(A), K, stride if nseg > 1 else rows, _lib.ptr(dm), nseg, ss, K, _lib.ptr(am), _lib.stream()), "hnr_absmax")
        outs.append((C, int(mx.item()), int(am.item())))
    Cs, Cp = outs[0][0], outs[1][0]
    for v in range(V):
        assert torch.equal(Cs[v * stride:v * stride + n], Cp[v * n:(v + 1) * n])
        assert bool((Cs[v * stride + n:(v + 1) * stride] == 7.0).all()), "rows between the segments were written"
    assert outs[0][1] == outs[1][1] and outs[0][2] == outs[1][2]
    assert np.frombuffer(np.int32(outs[0][2]).tobytes(), dtype=np.float32)[0] == float(packed.abs().max())
    # weight gradient over the segmented rows = over the packed rows (same 16-row blocks: n is not a multiple of 16, so the block boundaries
    # differ and only the fp64-referenced tolerance holds)
    res = []
    for Zt, Xt, nseg, ss, dm, rows in ((segz.to(dev), seg.to(dev), V, stride, dn, stride), (Zp.to(dev), packed.to(dev), 1, 0, dall, V * n)):
        mz = torch.zeros(1, dtype=torch.int32, device=dev); mx = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(L.hnr_absmax(_lib.ptr(Zt), N, rows, _lib.ptr(dm), nseg, ss, N, _lib.ptr(mz), _lib.stream()), "hnr_absmax")
        _lib.check(L.hnr_absmax(_lib.ptr(Xt), K, rows, _lib.ptr(dm), nseg, ss, K, _lib.ptr(mx), _lib.stream()), "hnr_absmax")
        scratch = torch.empty((int(L.hnr_h2wgrad_scratch_bytes(N, K)),), dtype=torch.uint8, device=dev)
        dW, db = torch.empty((N, K), device=dev), torch.empty((N,), device=dev)
        _lib.check(L.hnr_h2wgrad(_lib.ptr(Zt), N, _lib.ptr(Xt), K, rows, _lib.ptr(dm), nseg, ss, N, K, _lib.ptr(mz), _lib.ptr(mx), _lib.ptr(dW), K, _lib.ptr(db), 0,
                                 _lib.ptr(scratch), _lib.stream()), "hnr_h2wgrad")
        res.append((dW.cpu().numpy().astype(np.float64), db.cpu().numpy().astype(np.float64)))
    Z64, X64 = Zp.numpy().astype(np.float64), packed.numpy().astype(np.float64)
    ref, mag = Z64.T @ X64, np.abs(Z64).T @ np.abs(X64)
    for dW, db in res:
        assert (np.abs(dW - ref) / mag).max() < 4e-7
        assert (np.abs(db - Z64.sum(0)) / np.abs(Z64).sum(0)).max() < 4e-7


_WGRAD_AB = r'''
import sys, hashlib, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from hybridneuralrendering_amd import _lib
from tests.test_h2gemm_gpu import pack, _sign_words
L = _lib.lib()
dev = torch.device("cuda:0")
h = hashlib.sha1()
one = torch.tensor([np.float32(1.0).view(np.int32)], dtype=torch.int32, device=dev)
big = torch.tensor([np.float32(8.0).view(np.int32)], dtype=torch.int32, device=dev)
# weight gradients of the 256-wide shapes (ragged row counts, the ninth tile empty / partly full / full, padded row strides)
for M, N, K, ldz, ldx in ((40000, 256, 256, 256, 256), (33333, 256, 263, 264, 264), (777, 256, 287, 256, 288), (5, 256, 256, 264, 256)):
    g = torch.Generator(device="cpu").manual_seed(12 + M)
    Z, X = (torch.randn((M, ldz), generator=g) * 0.1).to(dev), torch.randn((M, ldx), generator=g).to(dev)
    scratch = torch.empty((int(L.hnr_h2wgrad_scratch_bytes(N, K)),), dtype=torch.uint8, device=dev)
    dW, db = torch.empty((N, K), device=dev), torch.empty((N,), device=dev)
    _lib.check(L.hnr_h2wgrad(_lib.ptr(Z), ldz, _lib.ptr(X), ldx, M, None, 1, 0, N, K, _lib.ptr(one), _lib.ptr(big), _lib.ptr(dW), K, _lib.ptr(db), 0, _lib.ptr(scratch), _lib.stream()), "hnr_h2wgrad")
    h.update(dW.cpu().numpy().tobytes()); h.update(db.cpu().numpy().tobytes())
# input gradients through the sign words (HNR_H2LIN_WS: weight-stationary / streaming kernel)
for M_cap, M, lda, ldc in ((20000 // 32 * 32, 19990, 256, 256), (4096, 4096, 264, 264), (64, 3, 256, 264)):
    g = torch.Generator(device="cpu").manual_seed(7 + M)
    A = (torch.randn((M_cap, lda), generator=g) * torch.exp(2.0 * torch.randn((M_cap, 1), generator=g))).to(dev)
    W = (torch.randn((256, 256), generator=g) / 16).to(dev)
    side = torch.randn((M_cap, 256), generator=g)
    img = pack([W], transposed=[True])[0]
    bits = torch.from_numpy(_sign_words(side.numpy(), M_cap)).to(dev)
    C = torch.full((M_cap, ldc), 7.0, device=dev)
    mx = torch.zeros(1, dtype=torch.int32, device=dev)
    dm = torch.tensor([M], dtype=torch.int64, device=dev)
    _lib.check(L.hnr_h2lin_dgrad_bits(_lib.ptr(A), lda, M_cap, _lib.ptr(dm), _lib.ptr(img), 256, 256, 0.01, _lib.ptr(bits), _lib.ptr(C), ldc, _lib.ptr(mx), _lib.stream()), "dgrad_bits")
    h.update(C.cpu().numpy().tobytes()); h.update(mx.cpu().numpy().tobytes())
print("WGRAD_SHA", h.hexdigest())
'''


def test_kernel_forms_of_the_256_wide_layers_agree_bit_for_bit(tmp_path):
    """The forms a 256-wide layer's gradients can run in give the same bits on the same seeded operands (the choices are read once per process: one
    process per form): weight gradient specialised for N = 256 / plain rows (HNR_WGRAD_DMA=1, the default) = the general DMA-staged kernel (=2) = the
    register-staged one (=0); input gradient weight-stationary (HNR_H2LIN_WS=1, the default) = streaming (=0)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text(_WGRAD_AB)
    sha = []
    for dma, ws in (("1", "1"), ("2", "0"), ("0", "1")):
        p = subprocess.run([sys.executable, str(script), root], capture_output=True, text=True, timeout=600, env=dict(os.environ, HNR_WGRAD_DMA=dma, HNR_H2LIN_WS=ws))
        assert p.returncode == 0, p.stderr[-2000:]
        sha.append([l for l in p.stdout.splitlines() if l.startswith("WGRAD_SHA")][0])
    assert sha[0] == sha[1] == sha[2], sha
