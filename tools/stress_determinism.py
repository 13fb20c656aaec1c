"""Is the forward path deterministic under GPU contention?  Renders the first scan-line block of the bench frame repeatedly while N other
processes keep the GPU busy (matmul loops / their own renders) and compares every render with the first one bit for bit.
python tools/stress_determinism.py [points] [iters] [hogs]"""
import sys, os, types, subprocess, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

if len(sys.argv) > 1 and sys.argv[1] == "--hog":
    dev = torch.device("cuda:0")
    if sys.argv[2] == "render":
        args = types.SimpleNamespace(scene="scene0241", points=2e5, width=640, height=480, margin=10)
        sc, opt, agg, cloud, rnd, cam = bench.build_world(args, dev, 1)
        t0 = time.time()
        while time.time() - t0 < float(sys.argv[3]):
            bench.render_frame(rnd, cloud, cam, sc, 0); torch.cuda.synchronize()
    else:
        a = torch.randn((4096, 4096), device=dev); t0 = time.time()
        while time.time() - t0 < float(sys.argv[3]):
            for _ in range(20): a = (a @ a).clamp(-1, 1)
            torch.cuda.synchronize()
    sys.exit(0)

args = types.SimpleNamespace(scene="scene0241", points=float(sys.argv[1]) if len(sys.argv) > 1 else 2e5, width=640, height=480, margin=10)
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
hogs = sys.argv[3].split(",") if len(sys.argv) > 3 else ["render"]
dev = torch.device("cuda:0")
sc, opt, agg, cloud, rnd, cam = bench.build_world(args, dev, 0)
R = cam["raydir"].shape[0]
half = dict(cam); half["raydir"] = cam["raydir"][:R // 2].contiguous()
col, out = bench.render_frame(rnd, cloud, half, sc, 0); torch.cuda.synchronize()
ref = {k: v.clone() for k, v in out.items() if torch.is_tensor(v)}
procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--hog", h, "40"]) for h in hogs if h != "none"]
time.sleep(8)
bad = 0
for it in range(iters):
    col, out = bench.render_frame(rnd, cloud, half, sc, 0); torch.cuda.synchronize()
    SR = out["sample_pidx"].shape[1]
    kept = torch.arange(SR, device=dev)[None, :] < out["ray_nsamp"][:, None].long()
    kept_ref = torch.arange(SR, device=dev)[None, :] < ref["ray_nsamp"][:, None].long()
    dq = dict(ray_nsamp=int((out["ray_nsamp"] != ref["ray_nsamp"]).sum()),
              pidx=int(((out["sample_pidx"] != ref["sample_pidx"]).any(dim=-1) & kept & kept_ref).sum()),
              loc=int(((out["sample_loc_w"] != ref["sample_loc_w"]).any(dim=-1) & kept & kept_ref).sum()),
              counts=int((out["counts"] != ref["counts"]).sum()))
    diffs = {k: int((v != ref[k]).sum()) for k, v in out.items() if torch.is_tensor(v) and k in ("coarse_raycolor", "coarse_point_opacity", "decoded", "ray_mask") and (v != ref[k]).any()}
    if diffs or any(dq.values()):
        bad += 1
        d = (out["decoded"] != ref["decoded"])
        idx = d.any(dim=-1).nonzero()
        comp = d.reshape(-1, 4).sum(dim=0).tolist()
        mag = float((out["decoded"] - ref["decoded"]).abs().max())
        print("iteration %d differs: query %s; %s; decoded components (sigma, r, g, b) differing %s, max |d| %.3e, first samples (ray, slot) %s" % (it, dq, diffs, comp, mag, idx[:6].tolist()))
        if idx.numel():
            r0, s0 = idx[0].tolist()
            print("   pidx now %s\n   pidx ref %s\n   decoded now %s ref %s" % (out["sample_pidx"][r0, s0].tolist(), ref["sample_pidx"][r0, s0].tolist(), out["decoded"][r0, s0].tolist(), ref["decoded"][r0, s0].tolist()))
print("%d of %d renders under contention (%s) differ from the quiet render" % (bad, iters, hogs))
for p in procs: p.wait()
