# round 5: busy fractions of the training step's kernels after the chain moved (texture addresser, VALU, MFMA, LDS, waiting)
cd $GRAFT_REPO_ROOT
HNR_PMC_CMD="$GRAFT_REPO_ROOT/tools/probe_train.py --steps 4" bash tools/run_pmc_frame.sh 2>&1 | grep -E "chain_ws_kernel|h2lin_kernel<16>|h2wgrad_dma|h2wgrad_kernel<8|segment_sum|ksum|mlp3|conv3x3_bwd_tile_kernel|upsample|extras|merge_bwd|chain_gather" | head -24
cp gpurun_out/r3f/frame_busy.json gpurun_out/r5_run40_train_busy.json
