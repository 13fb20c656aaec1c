# One rank's share of the patch-sharded C5 training step (49 dilated patches over N ranks: 7 or 6 patches each), rendered alone on ONE GPU, for every
# rank r of N (default 8): per-rank compute time + the bytes its collectives would move.  No scaling curve can be measured without the node; this is
# the per-rank floor.  bash tools/predict_train_scaling.sh [N] > profiles/r04_predict_train_8.txt
N=${1:-8}
cd $GRAFT_REPO_ROOT
echo "# whole batch on one GPU (49 patches)"
python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-f32-anchor --train-sharded-only 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1])['train_step_sharded']; print(json.dumps(d))"
for r in $(seq 0 $((N-1))); do
  HNR_BENCH_EMULATE_RANK=$r/$N python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-f32-anchor --train-sharded-only 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1])['train_step_sharded']; print(json.dumps({k:d[k] for k in ('emulated_rank','ms_per_step','compute_ms','valid_samples','neighbour_rows','collective_bytes')}))"
done
