"""Runs only the C5 patch-sharded training leg of bench.py (one rank's share under HNR_BENCH_EMULATE_RANK=r/n, or the whole batch) -- for rocprofv3
kernel traces / stats of the step.    python tools/probe_train_shard.py [--steps 6]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--warmup", type=int, default=2)
a = ap.parse_args()
sys.argv = [sys.argv[0]]
args = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
sc, opt, agg, cloud, rnd, cam = bench.build_world(args, dev, 0)
d = bench.train_leg_sharded(args, sc, opt, agg, cloud, rnd, cam, dev, 1, 0, False, os.environ.get("HNR_BENCH_EMULATE_RANK"), steps=a.steps, warmup=a.warmup)
print(json.dumps({k: d.get(k) for k in ("emulated_rank", "ms_per_step", "compute_ms", "allreduce_weights_ms", "exchange_points_ms", "step_form", "touched_points", "valid_samples", "error")}))
