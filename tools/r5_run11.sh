# round 5: full GPU suite + smoke + default bench line (the record run for this state of the tree)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run11; mkdir -p $O
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -6 $O/pytest_gpu.txt
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt
timeout 1500 python3 bench.py > $O/bench.txt 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_run11/bench.txt").read().strip().split("\n")[-1])
print("value", d["value"], "ms", d["ms_per_step"]); print("stage_ms", d["stage_ms"])
print("roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "mfma_busy", "avg_launch_ms")})
print("roofline_query", {k: d["roofline_query"].get(k) for k in ("achieved", "frac", "avg_launch_ms", "in_frame_ms", "kernel")})
for k in ("train_step", "train_step_sharded"):
    t = d.get(k) or {}
    print(k, {x: t.get(x) for x in ("ms_per_step", "captured_ms_per_step", "compute_ms", "step_form", "error")})
print("amortised", d["amortised_ms"]); print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
