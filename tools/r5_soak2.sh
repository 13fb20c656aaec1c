# round 5: the determinism soak again with the training chain in chain_ws_kernel<8> (counted vmcnt schedule incl. its stores) and the sign-word input gradients
cd $GRAFT_REPO_ROOT
D=gpurun_out/r5_soak2; mkdir -p $D
RACE_ITERS=6000 timeout 1200 python3 tools/race_c3.py > $D/race_default_quiet.txt 2>&1; echo "default side streams, quiet, 6000 steps: $(tail -1 $D/race_default_quiet.txt)"
timeout 600 python3 bench.py --steps 100000 --warmup 1 --no-cpu-baseline --no-train-leg --no-f32-anchor > /dev/null 2>&1 &
HOG=$!
sleep 30
RACE_ITERS=4000 timeout 900 python3 tools/race_c3.py > $D/race_default_contended.txt 2>&1; echo "default side streams, beside another process, 4000 steps: $(tail -1 $D/race_default_contended.txt)"
kill $HOG 2>/dev/null; wait $HOG 2>/dev/null
exit 0
