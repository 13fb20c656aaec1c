#!/bin/bash
# Determinism soak of the final tree (one gpurun call): contended renders, quiet and contended training steps -- every repeat must equal the first
# bit for bit.  gpurun -- 'bash tools/soak.sh <tag>'   (SOAK_RENDERS / SOAK_QUIET / SOAK_CONT: repeat counts, default 60 / 3000 / 2000)
cd "$GRAFT_REPO_ROOT" || exit 1
D=gpurun_out/${1:-soak}; mkdir -p "$D"
timeout 1500 python3 tools/stress_determinism.py 2e5 ${SOAK_RENDERS:-60} render,render > "$D/stress_render.txt" 2>&1; tail -1 "$D/stress_render.txt"
RACE_ITERS=${SOAK_QUIET:-3000} timeout 1500 python3 tools/race_c3.py > "$D/race_quiet.txt" 2>&1; echo "quiet, ${SOAK_QUIET:-3000} steps: $(tail -1 "$D/race_quiet.txt")"
timeout 1200 python3 bench.py --steps 100000 --warmup 1 --no-cpu-baseline --no-train-leg --no-f32-anchor > /dev/null 2>&1 &
HOG=$!
sleep 30
RACE_ITERS=${SOAK_CONT:-2000} timeout 1500 python3 tools/race_c3.py > "$D/race_contended.txt" 2>&1; echo "beside another process, ${SOAK_CONT:-2000} steps: $(tail -1 "$D/race_contended.txt")"
kill $HOG 2>/dev/null; wait $HOG 2>/dev/null
exit 0
