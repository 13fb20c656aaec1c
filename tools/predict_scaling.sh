# What each of N ranks would render, one after the other on ONE GPU (bench.py HNR_BENCH_EMULATE_RANK=r/N): the slowest rank's time bounds
# the strong-scaling step from below (the RCCL gather of 3.4 MB comes on top).  Usage: bash tools/predict_scaling.sh [N] [lines|blocks]
N=${1:-8}; MODE=${2:-lines}
OUT=$GRAFT_REPO_ROOT/gpurun_out/predict_${MODE}${BAND:-1}_$N.txt; : > $OUT
for r in $(seq 0 $((N-1))); do
  HNR_BENCH_EMULATE_RANK=$r/$N timeout 300 python3 bench.py --shard $MODE --band ${BAND:-1} --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('rank $r/$N $MODE: %.3f ms/step, %d rays, stages %s' % (d['ms_per_step'], d['config']['rays_per_step'], d['stage_ms']))" | tee -a $OUT
done
