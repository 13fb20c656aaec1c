"""Runs only the fwd+bwd training leg of bench.py -- for rocprofv3 kernel stats of the backward kernels.
    python tools/probe_train.py [--steps 10]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--points", type=float, default=2.0e6)
a = ap.parse_args()
sys.argv = [sys.argv[0], "--points", str(a.points)]
args = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
sc, opt, agg, cloud, rnd, cam = bench.build_world(args, dev, 0)
print(json.dumps(bench.train_leg(args, sc, opt, agg, cloud, rnd, cam, dev, steps=a.steps, warmup=2)))
