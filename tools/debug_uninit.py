"""Does the forward path read memory it never wrote?  Renders the first scan-line block of the bench frame from one process, with the caching
allocator's free memory filled with zeros, with NaN bit patterns and with large finite values before each render, and compares the colours
bit for bit.  python tools/debug_uninit.py [points]"""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

args = types.SimpleNamespace(scene="scene0241", points=float(sys.argv[1]) if len(sys.argv) > 1 else 2e5, width=640, height=480, margin=10)
dev = torch.device("cuda:0")
sc, opt, agg, cloud, rnd, cam = bench.build_world(args, dev, 0)
R = cam["raydir"].shape[0]
half = dict(cam); half["raydir"] = cam["raydir"][:R // 2].contiguous()

def poison(kind):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    n = min(int(free * 0.5), 24 << 30) // 4
    t = torch.empty((n,), dtype=torch.int32, device=dev)
    t.fill_(0 if kind == "zeros" else 0x7fc00001 if kind == "nan" else 0x7f000000)      # 0x7f000000 = 1.7e38
    del t                                            # back to the caching allocator: the next allocations reuse it un-cleared
    torch.cuda.synchronize()

outs = {}
for kind in ("zeros", "nan", "huge", "zeros"):
    poison(kind)
    col, out = bench.render_frame(rnd, cloud, half, sc, 0)
    torch.cuda.synchronize()
    c = col.cpu().numpy()
    if kind in outs:
        print("repeat %-5s: %d pixels differ from the first run with it" % (kind, int((c != outs[kind]).any(axis=1).sum())))
    else:
        outs[kind] = c
ref = outs["zeros"]
for kind in ("nan", "huge"):
    d = np.nonzero((outs[kind] != ref).any(axis=1))[0]
    print("free memory = %-5s: %d of %d pixels differ from the zero-filled run; first %s; max |d| %.3e; non-finite %d" % (
        kind, d.size, ref.shape[0], d[:8], float(np.nanmax(np.abs(outs[kind] - ref))) if d.size else 0.0, int((~np.isfinite(outs[kind])).sum())))
