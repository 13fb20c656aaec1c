#!/usr/bin/env python3
"""Headline benchmark: rays/s of the forward hybrid render (query -> gather/aggregate -> composite)
on the scene0241_01-like synthetic config (BASELINE.json configs[2] / SURVEY.md section 8d C3).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full 620x460 = 285 200-ray frame of the FIXED scene0241_01-like ray batch (north_star): with N ranks
the
frame's scan lines are dealt round-robin to the N ranks (parallel.shard_lines; --shard blocks: N contiguous blocks,
whose work differs
by up to 1.68x on this frame, tools/shard_balance.py), every rank renders its rays (cloud,
grid, weights and reference-view features replicated and already resident in HBM) and the colours are reassembled on
rank 0
with ONE RCCL gather -- strong scaling, value = 285 200 rays / max-over-ranks step time.  `--scaling weak` instead lets
every
rank render a whole frame of its own (value = N x 285 200 / time).

`python bench.py --gpus N` with WORLD_SIZE unset spawns the N ranks itself (child processes, before anything touches the
GPU);
under torch.distributed.run the launcher's RANK / LOCAL_RANK / WORLD_SIZE are used and must agree with --gpus.
Rank 0 prints ONE JSON line.
"""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from hnr_bench import (parse, spawn_ranks, build_world, render_frame, pmc_traffic, cpu_baseline,  # noqa: E402,F401
                       train_leg, train_leg_sharded, main)

if __name__ == "__main__":
    main()
