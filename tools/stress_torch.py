"""Control experiment: are plain PyTorch kernels deterministic on this box while other processes share the GPU?"""
import sys, os, subprocess, time
import torch
dev = torch.device("cuda:0")
torch.manual_seed(0)
a = torch.randn((64 << 20,), device=dev)
idx = torch.randint(0, a.numel(), (32 << 20,), device=dev)
def run():
    b = torch.sin(a) * 1.5 + a[idx].repeat(2)
    c = torch.cumsum(b.view(-1, 1024), dim=1)
    return b, c, (b > 0.5).nonzero().numel()
ref = run(); torch.cuda.synchronize()
here = os.path.dirname(os.path.abspath(__file__))
procs = [subprocess.Popen([sys.executable, os.path.join(here, "stress_determinism.py"), "--hog", h, "40"]) for h in ("render", "matmul")]
time.sleep(10)
bad = 0
for it in range(60):
    cur = run(); torch.cuda.synchronize()
    d0, d1 = int((cur[0] != ref[0]).sum()), int((cur[1] != ref[1]).sum())
    if d0 or d1 or cur[2] != ref[2]:
        bad += 1
        print("iteration %d: elementwise+gather %d, cumsum %d, nonzero count %d vs %d" % (it, d0, d1, cur[2], ref[2]))
print("%d of 60 torch runs differ under contention" % bad)
for p in procs: p.wait()
