"""A/B of the weight-stationary input-gradient kernel (csrc/h2lin_ws.hip, HNR_H2LIN_WS=1, the default) against the streaming kernel (=0):
   1. the launch alone at the C3 row count (HIP events), both forms, outputs and maximum word compared bit for bit, incl. a ragged row count;
   2. one golden-sized training step per form in a child process each: every output and gradient compared bit for bit;
   3. tools/probe_train.py per form: the step time and the block3 / block1 stage times.
python tools/ab_h2lin_ws.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child_kernel(out):
    import numpy as np
    import torch
    from hybridneuralrendering_amd import _lib
    from tests.test_h2gemm_gpu import pack
    import ctypes
    L = _lib.lib()
    dev = torch.device("cuda:0")
    fn = L.hnr_h2lin_dgrad_bits
    res = {}
    for M_cap, M in ((306944, 306936), (4096, 4001), (64, 33), (32, 32)):
        g = torch.Generator().manual_seed(M)
        Z = (torch.randn((M_cap, 264), generator=g) * torch.exp(2.0 * torch.randn((M_cap, 1), generator=g))).to(dev)
        W = (torch.randn((256, 256), generator=g) / 16).to(dev)
        img = pack([W], transposed=[True])[0]
        bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (M_cap // 32 * 4 * 64,), generator=g, dtype=torch.int64).to(torch.int32).to(dev)
        C = torch.full((M_cap, 264), 7.0, device=dev)
        mx = torch.zeros(1, dtype=torch.int32, device=dev)
        dm = torch.tensor([M], dtype=torch.int64, device=dev)
        def run():
            _lib.check(fn(_lib.ptr(Z), 264, M_cap, _lib.ptr(dm), _lib.ptr(img), 256, 256, 0.01, _lib.ptr(bits), _lib.ptr(C), 264, _lib.ptr(mx), _lib.stream()), "dgrad_bits")
        run(); torch.cuda.synchronize()
        res["C_%d" % M] = C.cpu().numpy(); res["mx_%d" % M] = mx.cpu().numpy()
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        print("HNR_H2LIN_WS=%s  M %6d : %.4f ms per launch" % (os.environ.get("HNR_H2LIN_WS"), M, e0.elapsed_time(e1) / 20), flush=True)
    np.savez(out, **res)


def child_step(out):
    sys.argv = [sys.argv[0], out]
    exec(open(os.path.join(ROOT, "tools", "ab_train_chain.py")).read(), {"__name__": "__main__", "__file__": os.path.join(ROOT, "tools", "ab_train_chain.py")})


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "kernel":
        child_kernel(sys.argv[2]); sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "step":
        child_step(sys.argv[2]); sys.exit(0)
    import numpy as np
    tmp = os.environ.get("TMPDIR", "/tmp")
    for what in ("kernel", "step"):
        files = []
        for ws in ("1", "0"):
            f = os.path.join(tmp, "ab_h2lin_%s_%s.npz" % (what, ws))
            env = dict(os.environ, HNR_H2LIN_WS=ws)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), what, f], env=env, capture_output=True, text=True)
            if what == "kernel" or r.returncode:
                print(r.stdout[-3000:], r.stderr[-3000:] if r.returncode else "")
            if r.returncode:
                sys.exit(1)
            files.append(f)
        a, b = np.load(files[0]), np.load(files[1])
        bad = [k for k in a.files if a[k].tobytes() != b[k].tobytes()]
        print("%s: %d arrays compared bit for bit, %d differ %s" % (what, len(a.files), len(bad), bad[:8]))
        for k in bad[:4]:
            d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
            print("   ", k, "max abs diff", d.max(), "of", np.abs(b[k]).max(), "count", int((d > 0).sum()))
    for ws in ("1", "0", "1", "0"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe_train.py"), "--steps", "20"], env=dict(os.environ, HNR_H2LIN_WS=ws, HNR_BENCH_TRAIN_GRAPH="0"),
                           capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print("HNR_H2LIN_WS=%s  step %.3f ms  bwd.block3 %.3f  bwd.block1 %.3f  fwd %.3f  bwd %.3f" % (ws, d["ms_per_step"], d["stage_ms"]["bwd.block3"], d["stage_ms"]["bwd.block1"],
                                                                                                   d["fwd_ms"], d["loss_bwd_ms"]))
        except Exception as ex:                                                   # noqa: BLE001
            print("probe_train failed", ex, r.stderr[-2000:])
