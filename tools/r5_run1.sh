# round 5, first GPU call: the graph-free / captured training step (tests, then the two training legs of bench.py and two emulated 1/8 shares)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run1; mkdir -p $O
timeout 900 python3 -m pytest tests/test_train_gpu.py -x -q -k "blur_module or captured" > $O/pytest_new.txt 2>&1; echo "pytest new rc=$?" >> $O/pytest_new.txt
tail -15 $O/pytest_new.txt
HNR_BENCH_STRICT=1 timeout 900 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32-anchor > $O/bench.txt 2> $O/bench.err; echo "bench rc=$?"
tail -3 $O/bench.err
python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r5_run1/bench.txt").read().strip().split("\n")[-1])
    print("frame ms", d["ms_per_step"], "value", d["value"])
    for k in ("train_step", "train_step_sharded"):
        t = d.get(k) or {}
        print(k, {x: t.get(x) for x in ("ms_per_step", "eager_ms_per_step", "compute_ms", "allreduce_weights_ms", "exchange_points_ms", "step_form", "error", "touched_points")})
except Exception as e:
    print("no bench line:", e)
PY
for r in 0 3; do
  HNR_BENCH_EMULATE_RANK=$r/8 timeout 600 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-f32-anchor --train-sharded-only 2>$O/emu$r.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1])['train_step_sharded']; print(json.dumps({k:d.get(k) for k in ('emulated_rank','ms_per_step','compute_ms','allreduce_weights_ms','exchange_points_ms','step_form','touched_points','error')}))" | tee $O/emu$r.txt
  HNR_BENCH_TRAIN_GRAPH=0 HNR_BENCH_EMULATE_RANK=$r/8 timeout 600 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-f32-anchor --train-sharded-only 2>>$O/emu$r.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1])['train_step_sharded']; print(json.dumps({k:d.get(k) for k in ('emulated_rank','ms_per_step','compute_ms','allreduce_weights_ms','exchange_points_ms','step_form','error')}))" | tee $O/emu${r}_eager.txt
done
