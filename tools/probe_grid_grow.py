"""Times hnr_grid_grow at the bench size: 2 M points, three successive grows of `add` points each (default 20 000 = 1 %), against a full rebuild.
python tools/probe_grid_grow.py [add] ; HNR_GRID_SLACK as for the build."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hybridneuralrendering_amd import scenes, _lib
from hybridneuralrendering_amd.querier import lighting_fast_querier, VoxelGrid
add = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
sc = scenes.make_scene("scene0241", 2000000, 2)
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
xyz = torch.from_numpy(sc.xyz).to(dev)
q = lighting_fast_querier(dev, sc.opt)
hp = q.get_hyperparameters(sc.opt.vsize, xyz[None], ranges=sc.opt.ranges)
g = VoxelGrid(xyz, hp[2][:3], hp[5], hp[6], sc.opt.query_size, sc.opt.P, sc.opt.max_o)
cur = xyz
for it in range(4):
    new = sc.xyz[rng.integers(0, sc.xyz.shape[0], size=add)] + rng.normal(0, 0.01, size=(add, 3)).astype(np.float32)
    new = np.clip(new, sc.xyz.min(0), sc.xyz.max(0))
    cur = torch.cat([cur, torch.from_numpy(new).to(dev)])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ok = g.grow(cur)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    print("grow %d: +%d points -> %s in %.3f ms (%s)" % (it, add, ok, ms, "" if ok else _lib.lib().hnr_last_error().decode()))
    if not ok:
        cur = cur[:-add]                      # refused (slack used up): nothing changed, the grid still describes the cloud without this batch
        break
torch.cuda.synchronize(); t0 = time.perf_counter()
g2 = VoxelGrid(cur, hp[2][:3], hp[5], hp[6], sc.opt.query_size, sc.opt.P, sc.opt.max_o)
torch.cuda.synchronize(); print("full rebuild of %d points: %.3f ms" % (cur.shape[0], (time.perf_counter() - t0) * 1e3))
for a, b in zip(g.export_runs() + g.export_dense(), g2.export_runs() + g2.export_dense()):
    assert torch.equal(a, b)
print("tables equal")
