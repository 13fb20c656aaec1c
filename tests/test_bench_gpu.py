"""bench.py contract: one JSON line with the fields the driver reads (small scene so the test takes seconds)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--points", "2e5",
                        "--cpu-sample-rays", "256"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "rays/s" and d["scaling"] == "strong" and d["dtype"] == "f32"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["rays_per_step"] * 1000.0 / d["ms_per_step"]) < 1e-3 * d["value"]
    assert "alpha_branch" in d["config"]["workload"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["value"] > 0 and c["psnr_gpu_vs_oracle_db"] > 60
    assert d["roofline_query"]["bound"] == "hbm" and d["train_step"]["ms_per_step"] > 0
    assert d["train_step_sharded"]["ms_per_step"] > 0 and d["train_step_sharded"]["n_ranks"] == 1 and d["train_step_sharded"]["collective_bytes"]["weights_allreduce"] > 1000000
    assert d["fp32_mfma_anchor"]["fp32_mfma_ms_per_step"] > 0 and d["fp32_mfma_anchor"]["max_abs_vs_f16x2_frame"] < 1e-4
    assert c["max_abs_gpu_vs_oracle_same_pixels"] <= 1e-4
    assert abs(r["frac_issued"] - r["achieved_issued"] / r["peak"]) < 1e-3 and r["frac"] < r["frac_issued"]


def test_bench_two_rank_rehearsal_of_the_patch_sharded_train_step():
    """BASELINE config 5 as it runs on N ranks (train_leg_sharded): two ranks on one GPU, gloo collectives on host copies -- control flow and shapes of
    the patch sharding + the two gradient collectives (one all-reduce of the flat weight gradients carrying the valid-ray counts, one fixed-capacity
    all-gather of packed point records), not a measurement."""
    e = dict(os.environ, HNR_BENCH_REHEARSAL="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--points", "2e5", "--no-cpu-baseline",
                        "--no-f32-anchor", "--train-sharded-only"], cwd=ROOT, capture_output=True, text=True, timeout=1500, env=e)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.strip().startswith("{")][0])
    t = d["train_step_sharded"]
    assert t.get("error") is None, t
    assert t["n_ranks"] == 2 and ("25 patches" in t["workload"] or "24 patches" in t["workload"])
    assert t["ms_per_step"] > 0 and t["touched_points"] > 0 and t["exchange_capacity"] >= t["touched_points"] and not t["exchange_overflow"]
    assert len(t["per_rank_ms_per_step"]) == 2 and t["collective_bytes"]["point_records_allgather_per_rank"] == (t["exchange_capacity"] + 2) * 160


def _run(extra, env=None, timeout=1500):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--points", "2e5", "--no-cpu-baseline",
                        "--no-train-leg"] + extra, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=e)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_gpus_2_spawns_two_ranks_and_reassembles_the_same_frame(tmp_path):
    """`python bench.py --gpus 2` with no launcher: the parent spawns the two ranks itself (rehearsal mode here: one GPU, gloo), the
    fixed frame's scan lines are dealt round-robin to the two ranks (default; `--shard blocks`: two contiguous blocks) and the gathered
    image equals the N=1 image (counterpart of the chunk scatter at /root/reference/run/test_ft.py:185-198)."""
    import numpy as np
    a, b, c = str(tmp_path / "n1.npy"), str(tmp_path / "n2.npy"), str(tmp_path / "n2b.npy")
    d1 = _run(["--gpus", "1", "--dump-colors", a])
    d2 = _run(["--gpus", "2", "--dump-colors", b], env={"HNR_BENCH_REHEARSAL": "1"})
    d3 = _run(["--gpus", "2", "--shard", "blocks", "--dump-colors", c], env={"HNR_BENCH_REHEARSAL": "1"})
    assert d1["n_gpus"] == 1 and d2["n_gpus"] == 2 and d2["scaling"] == "strong"
    assert "round-robin" in d2["config"]["parallelism"] and "contiguous" in d3["config"]["parallelism"]
    # both ways of dealing the rays are timed in the same run (the first N-GPU record settles SURVEY 8e's choice)
    for dd in (d2, d3):
        ab = dd["shard_ab"]
        assert ab["lines"]["ms_per_step_max_rank"] > 0 and ab["blocks"]["ms_per_step_max_rank"] > 0 and len(ab["lines"]["per_rank_ms"]) == 2
    assert d1["shard_ab"] is None and d1["rccl_ranks"] == 0
    assert np.array_equal(np.load(a), np.load(c))
    assert d2["config"]["rays_per_step"] == d1["config"]["rays_per_step"] and d2["config"]["rays_per_gpu"] * 2 == d1["config"]["rays_per_gpu"]
    assert "REHEARSAL" in d2["data"]
    c1, c2 = np.load(a), np.load(b)
    assert c1.shape == c2.shape == (d1["config"]["rays_per_step"], 3)
    diff = np.nonzero((c1 != c2).any(axis=1))[0]
    assert diff.size == 0, "%d of %d pixels differ between the 1-rank and the 2-rank frame (first %s, max |d| %.3e, block boundary at %d)" % (
        diff.size, c1.shape[0], diff[:8], float(np.abs(c1 - c2).max()), d2["config"]["rays_per_gpu"])


def test_bench_weak_scaling_flag_and_world_size_mismatch():
    d = _run(["--gpus", "2", "--scaling", "weak"], env={"HNR_BENCH_REHEARSAL": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["rays_per_step"] == 2 * d["config"]["rays_per_gpu"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_feature_map_repeats_bit_for_bit_beside_another_process():
    """The behaviour behind tests/test_abi_and_host.py::test_library_has_no_packed_instructions: hnr_image_features repeated while another process
    renders frames on the same GPU.  With packed fp32 instructions in featmap_kernel 38 % of such launches came out wrong in lanes 48..63
    (profiles/r04_contention.txt); 500 launches must all repeat the quiet result."""
    e = dict(os.environ, FM_ITERS="500", FM_HOG_WAIT="25")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "featmap_contention.py")], cwd=ROOT, capture_output=True, text=True, timeout=900, env=e)
    assert p.returncode == 0, p.stderr[-2000:]
    last = [l for l in p.stdout.splitlines() if "launches: feature map differs" in l][-1]
    assert last.startswith("0 of 500 launches"), last
