"""Per-phase cycle counts of the fused chain kernel (block 0), bench scene.  python tools/probe_chain.py [points]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import test_chain_gpu as T
from hybridneuralrendering_amd import _lib

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2000000
W = T._world(n_points=n, w=640, h=480, seed=2)
L, p = _lib.lib(), _lib.ptr
dev = W["dev"]
nv = W["n_valid"]
ws = torch.empty((int(L.hnr_chain_workspace_bytes(nv)),), dtype=torch.uint8, device=dev)
X5 = torch.empty((nv, 280), dtype=torch.float32, device=dev); sg = torch.empty((nv,), dtype=torch.float32, device=dev)
ptab = W["rnd"].point_table(W["cloud"]); q = W["q"]; c = W["cloud"]
# the product's sample list: hnr_chain_plan with the slot classes of the selected kernel (PROBE_CLASSES=0: the one-class list of hnr_sample_plan)
CL = min(int(os.environ.get("PROBE_CLASSES", "2")), int(L.hnr_chain_classes()))
if CL > 0:
    n_items = W["R"] * W["SR"]
    scratch = torch.empty((3 * ((n_items + 1023) // 1024) + 3,), dtype=torch.int32, device=dev)
    cnt = q["counts"].clone(); vs = torch.empty_like(W["vs_item"])
    _lib.check(L.hnr_chain_plan(p(q["work"]), p(q["sample_pidx"]), p(cnt), W["K"], n_items, CL, p(vs), nv, p(scratch), _lib.stream()), "plan")
    W["vs_item"] = vs; q = dict(q, counts=cnt); W["q"] = q
    print("classes %d: small %d tiny %d of %d" % (CL, int(cnt[_lib.CNT["SAMPLES_SMALL"]]), int(cnt[_lib.CNT["SAMPLES_TINY"]]), nv))
_lib.check(L.hnr_chain_gather(p(c.xyz), p(c.conf), p(c.dir), p(c.color), p(q["sample_pidx"]), p(q["sample_loc_w"]), p(W["raydir"]), p(W["campos"]),
                              p(W["camrot"]), p(W["vs_item"]), p(q["counts"]), W["SR"], W["K"], nv, p(ws), p(X5), 280, None, None, _lib.stream()), "g")
NW = 8 if os.environ.get('HNR_CHAIN_RT', '16') == '8' else 4          # waves per workgroup: 8 in the dual-group kernel (default)
for it in range(2):
    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g0.record()
    for _ in range(5):
        _lib.check(L.hnr_chain_gather(p(c.xyz), p(c.conf), p(c.dir), p(c.color), p(q["sample_pidx"]), p(q["sample_loc_w"]), p(W["raydir"]), p(W["campos"]),
                                      p(W["camrot"]), p(W["vs_item"]), p(q["counts"]), W["SR"], W["K"], nv, p(ws), p(X5), 280, None, None, _lib.stream()), "g")
    g1.record(); torch.cuda.synchronize()
print("chain_gather %.3f ms per launch" % (g0.elapsed_time(g1) / 5))
# the same gather reading hnr_point_records' 48-byte records: bit-identical workspace, fewer sectors per neighbour
rec = torch.empty((c.xyz.shape[0], 12), dtype=torch.float32, device=dev)
_lib.check(L.hnr_point_records(p(c.xyz), p(c.conf), p(c.dir), p(c.color), c.xyz.shape[0], p(rec), _lib.stream()), "r")
ws2 = torch.empty_like(ws); X5b = torch.empty_like(X5)
for it in range(2):
    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g0.record()
    for _ in range(5):
        _lib.check(L.hnr_chain_gather_rec(p(rec), p(q["sample_pidx"]), p(q["sample_loc_w"]), p(W["raydir"]), p(W["campos"]), p(W["camrot"]), p(W["vs_item"]),
                                          p(q["counts"]), W["SR"], W["K"], nv, p(ws2), p(X5b), 280, None, None, _lib.stream()), "g")
    g1.record(); torch.cuda.synchronize()
print("chain_gather_rec %.3f ms per launch; workspace identical: %s" % (g0.elapsed_time(g1) / 5, bool(torch.equal(ws, ws2))))
dbg = torch.zeros(((NW * 16 + 4 * 1024 + 64 + 16) * 2,), dtype=torch.float32, device=dev)
pk = W["agg"].packed_chain()
MODE = -int(os.environ.get('PROBE_MODE', '1'))          # -1: phase timing; -3 / -4 / -5 (dual-group kernel only): no epilogue work / same weights / both
for it in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(L.hnr_chain_forward(p(ws), p(ptab), 256, p(pk), p(q["counts"]), nv, 0.01, p(X5), 280, p(sg), p(dbg) if it == 2 else None, MODE, _lib.stream()), "f")
    e1.record(); torch.cuda.synchronize()
    print("chain_forward %.3f ms, %d valid samples, %d tiles" % (e0.elapsed_time(e1), nv, (nv + 15) // 16))
allv = dbg.view(torch.int64).cpu().numpy()
if os.environ.get('HNR_CHAIN_RT', '16') == '16':
    blk = allv[:4 * 1024].reshape(-1, 4); blk = blk[blk[:, 2] > 0]
    t = allv[4 * 1024:4 * 1024 + 64].reshape(4, 16)
    print('blocks %d: cycles/tile mean %.0f, GHz %.3f' % (len(blk), (blk[:, 0] / blk[:, 2]).mean(), (blk[:, 0] / blk[:, 1]).mean() * 0.1))
    for w in range(4):
        print('wave %d cycles per pass (L, rt): ' % w + ' '.join('%d' % (t[w, i] / max(blk[0, 2], 1)) for i in range(16)))
    f = allv[4 * 1024 + 64:4 * 1024 + 68]
    print('pass (0,1) of wave 0: first half %d, barrier wait %d, table-row load issue %d, second half %d' % tuple(int(x / max(blk[0, 2], 1)) for x in f))
    sys.exit(0)
t = allv[:NW * 16].reshape(NW, 16)
blk = allv[NW * 16:].reshape(-1, 4)
blk = blk[blk[:, 2] > 0]
ms = blk[:, 1] / 1e5
print('blocks %d: wall ms per block min %.2f mean %.2f max %.2f; tiles min %d max %d' % (len(blk), ms.min(), ms.mean(), ms.max(), blk[:, 2].min(), blk[:, 2].max()))
hw = blk[:, 3] >> 8
xcc = blk[:, 3] & 0xf
cu = (xcc << 8) | ((hw >> 8) & 0xff)               # (xcc, se_id[15:13], sh_id[12], cu_id[11:8])
from collections import defaultdict
by = defaultdict(list)
for i, c_ in enumerate(cu):
    by[int(c_)].append(i)
sizes = sorted(set(len(v) for v in by.values()))
pairs = [tuple(v) for v in by.values() if len(v) == 2][:6]
print('  distinct CUs %d, blocks per CU %s, example co-resident pairs %s' % (len(by), sizes, pairs))
d = [abs(v[1] - v[0]) for v in by.values() if len(v) == 2]
print('  |b1 - b0| of co-resident pairs: %s' % sorted(set(d))[:10])
fast = [min(ms[v[0]], ms[v[1]]) for v in by.values() if len(v) == 2]; slow = [max(ms[v[0]], ms[v[1]]) for v in by.values() if len(v) == 2]
if fast: print('  per CU: first finisher mean %.2f ms, second mean %.2f ms' % (np.mean(fast), np.mean(slow)))
for x in range(8):
    m = (np.arange(len(blk)) % 8) == x
    print('  blocks b%%8==%d: xcc %s, ms mean %.2f max %.2f, GHz %.3f' % (x, sorted(set(xcc[m].tolist())), ms[m].mean(), ms[m].max(), (blk[m, 0] / blk[m, 1]).mean() * 0.1))
names = ["prologue", "L0 mfma", "L0 act(+T wait)", "L0 publish", "L1 mfma", "L1/L2 act", "L1/L2 publish", "L2 mfma", "L3 mfma", "L3 epilogue"]
for w in range(NW):
    nt = max(t[w, 12], 1)
    print("wave %d: %d tiles, %.0f cycles/tile (%.3f GHz): " % (w, nt, t[w, 10] / nt, t[w, 10] / max(t[w, 11], 1) * 0.1) + ", ".join("%s %.0f" % (names[i], t[w, i] / nt) for i in range(10)))
