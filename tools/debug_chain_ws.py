"""Where does a chain kernel variant differ from the fp64 evaluation?  HNR_CHAIN_RT=16 python tools/debug_chain_ws.py [layer]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import test_chain_gpu as T

layer = int(sys.argv[1]) if len(sys.argv) > 1 else 0
W = T._world()
Xd, ext, wagg, row_pid, Cbuf = T._per_layer_inputs(W)
prow = T._padded_rows(W)
Hs64, X564, sig64 = T._fp64_chain(W, Xd, ext, wagg, row_pid, T._run_chain(W)[3])
_, _, dbg, _ = T._run_chain(W, dbg_layer=layer)
got = dbg[prow].double()
ref = Hs64[layer].double()
err = (got - ref).abs() / ref.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
bad = err > 1e-4
print("rows %d, bad entries %d of %d" % (got.shape[0], int(bad.sum()), bad.numel()))
pr = prow.cpu().numpy()
badr = bad.any(dim=1).cpu().numpy()
print("bad rows %d; by row tile (padded row %% 128 // 32): %s" % (badr.sum(), np.bincount((pr[badr] % 128) // 32, minlength=4)))
print("bad rows by tile index (first 20 tiles): %s" % np.bincount(pr[badr] // 128)[:20])
badc = bad.any(dim=0).cpu().numpy()
print("bad columns by 16-column group: %s" % badc.reshape(16, 16).sum(axis=1))
if badr.any():
    r = int(np.nonzero(badr)[0][0])
    print("first bad row %d (padded %d): got %s\n ref %s" % (r, pr[r], got[r, :8].cpu().numpy(), ref[r, :8].cpu().numpy()))
X5c, sigc, _, _ = T._run_chain(W)
print("X5 max rel err %.3e, sigma %.3e" % (T._rel(X5c[:, :256], X564), T._rel(sigc, sig64)))
e5 = (X5c[:, :256].double() - X564.double()).abs() / X564.double().abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
b5 = e5 > 1e-4
print("X5 bad entries %d of %d; bad columns by 4-column group: %s" % (int(b5.sum()), b5.numel(), b5.any(dim=0).cpu().numpy().reshape(64, 4).sum(axis=1)))
rows = b5.any(dim=1).cpu().numpy()
print("X5 bad samples %d; by sample %% 16: %s" % (rows.sum(), np.bincount(np.nonzero(rows)[0] % 16, minlength=16)))
if rows.any():
    r = int(np.nonzero(rows)[0][0])
    print("sample %d got %s\n ref %s" % (r, X5c[r, :12].cpu().numpy(), X564[r, :12].cpu().numpy()))
if rows.any():
    r = int(np.nonzero(rows)[0][0])
    cols = np.nonzero(b5[r].cpu().numpy())[0]
    print("sample %d bad columns %s" % (r, cols[:40]))
    print(" got %s\n ref %s" % (X5c[r, cols[:8]].cpu().numpy(), X564[r, cols[:8]].cpu().numpy()))
    # is the wrong value another column's right value?
    g = X5c[r, :256].double().cpu().numpy(); rf = X564[r].double().cpu().numpy()
    for cidx in cols[:6]:
        near = np.argmin(np.abs(rf - g[cidx]))
        print("  col %d got %.6g = ref of col %d (%.6g)?" % (cidx, g[cidx], near, rf[near]))
