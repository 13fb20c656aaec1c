import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from tests import test_chain_gpu as T
W = T._world(n_points=20000, w=640, h=480, seed=3)
sc = W["sc"]; rnd = W["rnd"]
img = torch.from_numpy(np.ascontiguousarray(sc.images_nearest)).to(W["dev"])
fm = rnd.feature_map(img)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    rnd._fm_key = None; fm = rnd.feature_map(img)
e1.record(); torch.cuda.synchronize()
print("feature map %.3f ms" % (e0.elapsed_time(e1) / 10), tuple(fm.shape))
np.save(sys.argv[1], fm.cpu().numpy())
