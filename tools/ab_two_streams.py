"""A/B: K frames of the bench workload rendered back to back on ONE stream vs alternating on TWO streams (each frame still one hnr_render_forward with
its own workspace): does the front of frame n+1 (query, gather: HBM-bound) overlap with the back of frame n (per-sample MLPs: issue-bound)?
python tools/ab_two_streams.py [frames]"""
import os, sys, time, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from hybridneuralrendering_amd.render import HybridRenderer

args = types.SimpleNamespace(scene="scene0241", points=2000000, width=640, height=480, margin=10, knn_order=None)
dev = torch.device("cuda:0")
sc, opt, agg, cloud, rnd, cam = bench.build_world(args, dev, 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rnd2 = HybridRenderer(opt, agg, dev)                      # its own caches / keep-alive slots (the grid is rebuilt once: same tables)
rnds = [rnd, rnd2]
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
for r in rnds:                                            # warm: grid, tables, feature map
    bench.render_frame(r, cloud, cam, sc, 0)
torch.cuda.synchronize()
ref = bench.render_frame(rnd, cloud, cam, sc, 0)[0].clone()


def run(two):
    cols = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K):
        s = streams[i & 1] if two else streams[0]
        r = rnds[i & 1] if two else rnds[0]
        with torch.cuda.stream(s):
            r._fm_key = None                              # a new frame has new reference views (as in bench.py)
            col, _ = bench.render_frame(r, cloud, cam, sc, 0)
        cols.append(col)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    assert all(torch.equal(c, ref) for c in cols)
    return dt


for rep in range(3):
    a, b = run(False), run(True)
    print("rep %d: one stream %.3f ms/frame, two streams %.3f ms/frame (%.1f %%)" % (rep, a, b, 100.0 * (a - b) / a))
