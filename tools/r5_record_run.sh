# round 5 record run: full GPU suite, smoke, default bench line, bench kernel stats
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run23; mkdir -p $O
timeout 3000 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt
timeout 1500 python3 bench.py > $O/bench.txt 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_run23/bench.txt").read().strip().split("\n")[-1])
print("value", d["value"], "ms", d["ms_per_step"]); print("stage_ms", d["stage_ms"])
print("roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "mfma_busy", "avg_launch_ms", "traffic")})
print("roofline_query", {k: d["roofline_query"].get(k) for k in ("achieved", "frac", "avg_launch_ms", "in_frame_ms", "traffic")})
for k in ("train_step", "train_step_sharded"):
    t = d.get(k) or {}
    print(k, {x: t.get(x) for x in ("ms_per_step", "captured_ms_per_step", "compute_ms", "step_form", "error")})
print("amortised", d["amortised_ms"]); print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"].get("rays_gathering_another_pixel"))
PY
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/st5
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st5 -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-train-leg --steps 4 --warmup 1 > /tmp/st5.log 2>&1
cp /tmp/st5/*kernel_stats.csv $GRAFT_REPO_ROOT/$O/bench_kernel_stats.csv 2>/dev/null
