"""Which stage of the per-neighbour chain is not deterministic under GPU contention?  Runs hnr_chain_gather + hnr_chain_forward on one fixed
set of query outputs repeatedly while other processes keep the GPU busy, and compares the gather's workspace (layer-0 images + row scalars),
X5 and sigma with a quiet reference run, bit for bit.   python tools/stress_chain.py [points] [iters]"""
import sys, os, subprocess, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import test_chain_gpu as T
from hybridneuralrendering_amd import _lib

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
W = T._world(n_points=n, w=640, h=480, seed=2)
L, p, dev, cloud, q, rnd = _lib.lib(), _lib.ptr, W["dev"], W["cloud"], W["q"], W["rnd"]
nv = W["n_valid"]
ptab = rnd.point_table(cloud)
pk = W["agg"].packed_chain()

QUIET_WS = None
def run():
    ws = torch.zeros((int(L.hnr_chain_workspace_bytes(nv)),), dtype=torch.uint8, device=dev)
    X5 = torch.zeros((nv, 280), dtype=torch.float32, device=dev); sg = torch.zeros((nv,), dtype=torch.float32, device=dev)
    _lib.check(L.hnr_chain_gather(p(cloud.xyz), p(cloud.conf), p(cloud.dir), p(cloud.color), p(q["sample_pidx"]), p(q["sample_loc_w"]), p(W["raydir"]), p(W["campos"]),
                                  p(W["camrot"]), p(W["vs_item"]), p(q["counts"]), W["SR"], W["K"], nv, p(ws), p(X5), 280, None, None, _lib.stream()), "g")
    torch.cuda.synchronize()
    if QUIET_WS is not None: ws.copy_(QUIET_WS)              # forward-only test: the gather's output of the quiet run
    ws_g = ws.clone()
    _lib.check(L.hnr_chain_forward(p(ws), p(ptab), 256, p(pk), p(q["counts"]), nv, 0.01, p(X5), 280, p(sg), None, 0, _lib.stream()), "f")
    torch.cuda.synchronize()
    return ws_g, ws.clone(), X5, sg

ref = run()
QUIET_WS = ref[0].clone() if os.environ.get('STRESS_FORWARD_ONLY') else None
here = os.path.dirname(os.path.abspath(__file__))
procs = [subprocess.Popen([sys.executable, os.path.join(here, "stress_determinism.py"), "--hog", h, "45"]) for h in ("render", "matmul")]
time.sleep(10)
bad = 0
for it in range(iters):
    cur = run()
    names = ["workspace after gather", "workspace after forward", "X5", "sigma"]
    d = [int((a != b).sum()) for a, b in zip(cur, ref)]
    if any(d):
        bad += 1
        msg = "; ".join("%s: %d" % (nm, k) for nm, k in zip(names, d) if k)
        extra = ""
        if d[0]:
            idx = np.nonzero(cur[0].cpu().numpy() != ref[0].cpu().numpy())[0]
            groups = 4 * ((nv + 15) // 16)
            xp_bytes = groups * 8192
            inxp = idx[idx < xp_bytes]; inaux = idx[idx >= xp_bytes] - xp_bytes
            a16, b16 = cur[0].cpu().numpy().view(np.uint16), ref[0].cpu().numpy().view(np.uint16)
            w = np.nonzero(a16 != b16)[0][:12]
            print('   fp16 words now %s\n   fp16 words ref %s\n   plane of the word (0 = h, 1 = m): %s' % ([hex(x) for x in a16[w]], [hex(x) for x in b16[w]], ((w * 2 % 2048) // 1024).tolist()))
            extra = " | gather diffs: %d bytes in xp (groups %s, byte-in-group %s), %d in aux (groups %s, byte-in-group %s)" % (
                inxp.size, sorted(set((inxp // 8192).tolist()))[:6], sorted(set((inxp % 8192).tolist()))[:6], inaux.size, sorted(set((inaux // 1280).tolist()))[:6], sorted(set((inaux % 1280).tolist()))[:12])
        if d[2]:
            rows = np.nonzero((cur[2].cpu().numpy() != ref[2].cpu().numpy()).any(axis=1))[0]
            extra += " | X5 samples %s (groups %s; n_valid %d)" % (rows[:10].tolist(), sorted(set((rows // 4).tolist()))[:6], nv)
        print("iteration %d: %s%s" % (it, msg, extra))
print("%d of %d runs differ" % (bad, iters))
for pr in procs: pr.wait()
