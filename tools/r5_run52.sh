# round 5: emulated shares of the strong-scaling frame on one GPU (8 / 4 / 2 ranks, scan lines dealt round-robin) + the whole frame, same box
cd $GRAFT_REPO_ROOT
bash tools/predict_scaling.sh 8 lines > /dev/null 2>&1
for n in 4 2; do for r in 0 $((n-1)); do
  HNR_BENCH_EMULATE_RANK=$r/$n timeout 300 python3 bench.py --shard lines --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('rank $r/$n lines: %.3f ms/step, %d rays' % (d['ms_per_step'], d['config']['rays_per_step']))" >> gpurun_out/predict_lines1_8.txt
done; done
timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('whole frame: %.3f ms/step, %d rays' % (d['ms_per_step'], d['config']['rays_per_step']))" >> gpurun_out/predict_lines1_8.txt
cat gpurun_out/predict_lines1_8.txt | cut -c1-80
