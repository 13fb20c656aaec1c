cd $GRAFT_REPO_ROOT
HNR_BENCH_TRAIN_GRAPH=0 HNR_PMC_CMD="$GRAFT_REPO_ROOT/tools/probe_train.py --steps 4" bash tools/run_pmc_frame.sh 2>&1 | grep -E "chain_kernel|h2lin_kernel<16>|h2wgrad_dma|segment_sum|ksum|mlp3|conv3x3_bwd_tile_kernel<24|upsample" | head -20
cp gpurun_out/r3f/frame_busy.json gpurun_out/r5_run25_train_busy.json
