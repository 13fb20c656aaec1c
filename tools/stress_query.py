"""Randomised bit-exactness sweep of the HIP query (grid build + march + k-NN, all kernel variants) against the C oracle.
Not a test (minutes of GPU time); run occasionally:  python tools/stress_query.py [n_cases]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_query_gpu as T  # noqa: E402
from hybridneuralrendering_amd import querier as Q  # noqa: E402
from oracle import query_oracle as qo  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(2024)
bad = 0
for i in range(n_cases):
    K = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 8, 8, 8, 12, 16]))
    cs = T._case(seed=int(rng.integers(1, 10 ** 6)), n=int(rng.choice([300, 3000, 20000, 80000])), P=int(rng.choice([1, 2, 5, 9, 26, 30, 63])),
                 max_o=int(rng.choice([50, 2000, 100000, 610000])), K=K, SR=int(rng.choice([1, 8, 24, 40, 80])),
                 R=int(rng.choice([1, 77, 1500, 3072])), D=int(rng.choice([64, 200, 400])),
                 size=tuple(rng.uniform(0.3, 1.2, size=3)), near=0.05, far=float(rng.uniform(0.4, 2.0)))
    tm = cs["tmid"]
    if rng.random() < 0.3:                                   # per-ray jittered depths
        tm = (tm[None].repeat(cs["rays"].shape[0], 0) * (1 + 0.01 * rng.uniform(-1, 1, size=(cs["rays"].shape[0], tm.shape[0])))).astype(np.float32)
    try:
        og, ref, g, res = T._run_both(cs, tmid=tm)
        T._assert_query_equal(ref, res, cs)
    except AssertionError as e:
        bad += 1
        print("MISMATCH case %d: K=%d P=%d max_o=%d SR=%d R=%d n=%d: %s" % (i, K, cs["P"], cs["max_o"], cs["SR"], cs["rays"].shape[0], cs["xyz"].shape[0],
                                                                             str(e)[:300]))
    finally:
        torch.cuda.synchronize()
print("stress_query: %d cases, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
