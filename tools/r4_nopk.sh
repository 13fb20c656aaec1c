D=gpurun_out/r4_nopk; mkdir -p $D
timeout 2400 python -m pytest tests -m gpu -x -q > $D/pytest_gpu.txt 2>&1; tail -3 $D/pytest_gpu.txt
timeout 900 python bench.py > $D/bench.json 2> $D/bench.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4_nopk/bench.json') if l.strip().startswith('{')][0])
print(d['value'], d['ms_per_step'], d['stage_ms'], 'train', d['train_step']['ms_per_step'], 'C5', d['train_step_sharded']['ms_per_step'])
PY
FM_ITERS=2000 python tools/featmap_contention.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee $D/featmap.txt
(timeout 500 python bench.py --steps 100000 --warmup 1 --no-cpu-baseline --no-train-leg --no-f32-anchor > /dev/null 2>&1 &) ; sleep 30
for sd in 0 15 31; do HNR_TRAIN_SIDE=$sd RACE_ITERS=1500 timeout 600 python tools/race_c3.py > $D/race_contended_side$sd.txt 2>&1; echo "side $sd contended: $(tail -1 $D/race_contended_side$sd.txt)"; done
