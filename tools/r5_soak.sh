# round 5: the round-4 soak with the round-5 kernels (query over neighbourhood lists, wave-path per-point sums, blur select, exchange): quiet, contended, and the frame under contention
cd $GRAFT_REPO_ROOT
D=gpurun_out/r5_soak; mkdir -p $D
RACE_ITERS=10000 timeout 1200 python3 tools/race_c3.py > $D/race_default_quiet.txt 2>&1; echo "default side streams, quiet, 10000 steps: $(tail -1 $D/race_default_quiet.txt)"
timeout 900 python3 tools/stress_determinism.py 2e5 120 render,render > $D/stress_render.txt 2>&1; tail -1 $D/stress_render.txt
timeout 600 python3 bench.py --steps 100000 --warmup 1 --no-cpu-baseline --no-train-leg --no-f32-anchor > /dev/null 2>&1 &
HOG=$!
sleep 30
RACE_ITERS=5000 timeout 900 python3 tools/race_c3.py > $D/race_default_contended.txt 2>&1; echo "default side streams, beside another process, 5000 steps: $(tail -1 $D/race_default_contended.txt)"
kill $HOG 2>/dev/null; wait $HOG 2>/dev/null
exit 0
