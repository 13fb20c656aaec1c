cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run33; mkdir -p $O
HNR_TRAIN_CHAIN_WS=1 timeout 600 python3 tools/ab_train_chain.py /tmp/ws1.npz > $O/ws1.txt 2>&1
HNR_TRAIN_CHAIN_WS=0 timeout 600 python3 tools/ab_train_chain.py /tmp/ws0.npz > $O/ws0.txt 2>&1
python3 - <<'PY' > $O/cmp.txt 2>&1
import numpy as np
a, b = np.load("/tmp/ws1.npz"), np.load("/tmp/ws0.npz")
for k in a.files:
    x, y = a[k].astype(np.float64), b[k].astype(np.float64)
    if x.shape != y.shape: print(k, "SHAPE", x.shape, y.shape); continue
    sc = max(np.abs(y).max(), 1e-30)
    print("%-50s max|ref| %.3e  max diff / max|ref| %.2e  equal %s" % (k, sc, np.abs(x - y).max() / sc, np.array_equal(a[k], b[k])))
PY
cat $O/cmp.txt
