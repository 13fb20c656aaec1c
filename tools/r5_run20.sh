# round 5: determinism soak of the training step with the round-5 kernels, per-kernel busy fractions of the frame and of the one-lane k-NN
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run20; mkdir -p $O
RACE_ITERS=2000 timeout 900 python3 tools/race_c3.py 2>&1 | tail -2 | tee $O/race_c3.txt
bash tools/run_pmc_frame.sh > $O/frame_busy.txt 2>&1; tail -25 $O/frame_busy.txt
cp gpurun_out/r3f/frame_busy.json $O/frame_busy.json 2>/dev/null
HNR_PMC_CMD="$GRAFT_REPO_ROOT/tools/probe_query.py" PROBE_PAD=0 PROBE_KNN_ORDER=1 HNR_KNN=7 bash tools/run_pmc_frame.sh > $O/query7_busy.txt 2>&1; tail -6 $O/query7_busy.txt
cp gpurun_out/r3f/frame_busy.json $O/query7_busy.json 2>/dev/null
HNR_PMC_CMD="$GRAFT_REPO_ROOT/tools/probe_query.py" PROBE_PAD=0 PROBE_KNN_ORDER=1 HNR_KNN=8 bash tools/run_pmc_frame.sh > $O/query8_busy.txt 2>&1; tail -6 $O/query8_busy.txt
cp gpurun_out/r3f/frame_busy.json $O/query8_busy.json 2>/dev/null
