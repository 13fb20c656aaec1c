"""hnr_h2wgrad 256 x 256 at the C3 batch's row count: the kernel specialised for the 256-wide layers (HNR_WGRAD_DMA=1, default) vs the general DMA-staged
kernel (=2) vs the register-staged one (=0).  The switch is read once per process: `python tools/ab_wgrad.py` runs itself once per form and compares
the results bit for bit; `python tools/ab_wgrad.py child <out.npz>` is one form."""
import os, sys, subprocess
import numpy as np
if len(sys.argv) < 2:
    outs = {}
    for mode in ("1", "2", "0", "1", "2"):
        f = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ab_wgrad_%s.npz" % mode)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", f], env=dict(os.environ, HNR_WGRAD_DMA=mode), capture_output=True, text=True)
        print(r.stdout.strip() if r.returncode == 0 else r.stderr[-2000:])
        outs[mode] = np.load(f)
    for m in ("2", "0"):
        bad = [k for k in outs["1"].files if outs["1"][k].tobytes() != outs[m][k].tobytes()]
        print("HNR_WGRAD_DMA=1 vs %s: %d arrays compared bit for bit, %d differ %s" % (m, len(outs["1"].files), len(bad), bad))
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridneuralrendering_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
torch.manual_seed(0)                 # seeded operands: the checksums of the two runs are comparable (tests/test_h2gemm_gpu.py holds the bit-for-bit test)
res = {}
for (M, N, K) in ((306832, 256, 256), (306832, 256, 263), (306821, 256, 256), (1000, 256, 287), (306832, 256, 60), (306821, 256, 60), (37, 256, 63)):
    Z, X = torch.randn((M, N), device=dev), torch.randn((M, K + (4 - K % 4) % 4), device=dev)
    mz = torch.tensor([np.float32(8.0).view(np.int32)], dtype=torch.int32, device=dev)
    scratch = torch.empty((int(L.hnr_h2wgrad_scratch_bytes(N, K)),), dtype=torch.uint8, device=dev)
    dW, db = torch.empty((N, K), device=dev), torch.empty((N,), device=dev)
    run = lambda: _lib.check(L.hnr_h2wgrad(_lib.ptr(Z), N, _lib.ptr(X), X.shape[1], M, None, 1, 0, N, K, _lib.ptr(mz), _lib.ptr(mz), _lib.ptr(dW), K, _lib.ptr(db), 0, _lib.ptr(scratch), _lib.stream()), "wgrad")
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    ref = Z.double().t() @ X[:, :K].double()
    err = float((dW.double() - ref).abs().max() / ref.abs().max())
    print("HNR_WGRAD_DMA=%s M=%d N=%d K=%d: %.4f ms per launch (+ reduce), %.2f TB/s of operands, max rel err vs fp64 %.2e, checksum %.9e" % (
        os.environ.get("HNR_WGRAD_DMA", "1"), M, N, K, ms, M * (N + K) * 4 / ms / 1e9, err, float(dW.double().sum())))
    res["dW_%d_%d" % (M, K)] = dW.cpu().numpy(); res["db_%d_%d" % (M, K)] = db.cpu().numpy()
np.savez(sys.argv[2], **res)
