"""A/B timing of chain kernel builds in ONE process: python tools/ab_chain.py name=path.so [name=path.so ...] [--rounds N]
Every variant runs hnr_chain_forward (product mode) on the same bench-frame workspace, interleaved round by round; prints per-variant
median / min ms and checks that all variants produce identical X5 / sigma bits as the first one (or reports the max difference)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import test_chain_gpu as T
from hybridneuralrendering_amd import _lib

rounds = 7
vs = []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--rounds": rounds = int(args.pop(0))
    else: n, pth = a.split("=", 1); vs.append((n, os.path.abspath(pth)))
n_pts = 2000000
W = T._world(n_points=n_pts, w=640, h=480, seed=2)
L, p = _lib.lib(), _lib.ptr
dev = W["dev"]; nv = W["n_valid"]
ws = torch.empty((int(L.hnr_chain_workspace_bytes(nv)),), dtype=torch.uint8, device=dev)
X5 = torch.empty((nv, 280), dtype=torch.float32, device=dev); sg = torch.empty((nv,), dtype=torch.float32, device=dev)
ptab = W["rnd"].point_table(W["cloud"]); q = W["q"]; c = W["cloud"]
n_items = W["R"] * W["SR"]
scratch = torch.empty((3 * ((n_items + 1023) // 1024) + 3,), dtype=torch.int32, device=dev)
cnt = q["counts"].clone(); vsi = torch.empty_like(W["vs_item"])
_lib.check(L.hnr_chain_plan(p(q["work"]), p(q["sample_pidx"]), p(cnt), W["K"], n_items, 2, p(vsi), nv, p(scratch), _lib.stream()), "plan")
rec = torch.empty((c.xyz.shape[0], 12), dtype=torch.float32, device=dev)
_lib.check(L.hnr_point_records(p(c.xyz), p(c.conf), p(c.dir), p(c.color), c.xyz.shape[0], p(rec), _lib.stream()), "r")
_lib.check(L.hnr_chain_gather_rec(p(rec), p(q["sample_pidx"]), p(q["sample_loc_w"]), p(W["raydir"]), p(W["campos"]), p(W["camrot"]), p(vsi),
                                  p(cnt), W["SR"], W["K"], nv, p(ws), p(X5), 280, None, None, _lib.stream()), "g")
pk = W["agg"].packed_chain()
torch.cuda.synchronize()
libs = []
for n, pth in vs:
    l = ctypes.CDLL(pth)
    l.hnr_chain_forward.restype = ctypes.c_int
    l.hnr_chain_forward.argtypes = L.hnr_chain_forward.argtypes
    libs.append((n, l))
def run(l, X, S):
    return l.hnr_chain_forward(p(ws), p(ptab), 256, p(pk), p(cnt), nv, 0.01, p(X), 280, p(S), None, 0, _lib.stream())
ref = None
times = {n: [] for n, _ in libs}
for r in range(rounds + 1):
    for n, l in libs:
        Xo = torch.zeros_like(X5); So = torch.zeros_like(sg)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rc = run(l, Xo, So); e1.record(); torch.cuda.synchronize()
        assert rc == 0, (n, rc)
        if r > 0: times[n].append(e0.elapsed_time(e1))
        if r == 0:
            if ref is None: ref = (Xo[:, :256].clone(), So.clone())
            else:
                same = torch.equal(Xo[:, :256], ref[0]) and torch.equal(So, ref[1])
                dx = float((Xo[:, :256] - ref[0]).abs().max()); ds = float((So - ref[1]).abs().max())
                print("%s vs %s: bit-identical %s (max |dX5| %.3e of %.3e, max |dsigma| %.3e)" % (n, libs[0][0], same, dx, float(ref[0].abs().max()), ds))
for n, _ in libs:
    t = np.array(times[n]); print("%-12s median %.3f ms  min %.3f  max %.3f  (%d rounds)" % (n, np.median(t), t.min(), t.max(), len(t)))
