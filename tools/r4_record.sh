# Round-4 A/B records as files (the runs of tools/r4_cf.sh / r4_wg.sh printed to the terminal only) + the neighbour-count histogram of the bench frame.
D=gpurun_out/${1:-r4_record}; mkdir -p $D
python tools/nb_hist.py 2>/dev/null | tail -2 > $D/nb_hist.txt; cat $D/nb_hist.txt
{ echo "# tools/ab_cf.py: colour-feature MLP on the bench frame's rows, HNR_CF_WS=1 (cf_ws_kernel) vs 0 (mlp3_kernel), two processes each";
  for i in 1 2; do HNR_CF_WS=1 timeout 600 python tools/ab_cf.py 2>&1 | grep HNR_; HNR_CF_WS=0 timeout 600 python tools/ab_cf.py 2>&1 | grep HNR_; done; } > $D/r04_cf_ws_ab.txt; cat $D/r04_cf_ws_ab.txt
{ echo "# tools/ab_wgrad.py: 256x256 weight gradient at 306 k rows, HNR_WGRAD_DMA=1 (h2wgrad_dma_kernel, default) vs 0";
  for i in 1 2; do python tools/ab_wgrad.py 2>&1 | grep HNR_; HNR_WGRAD_DMA=0 python tools/ab_wgrad.py 2>&1 | grep HNR_; done;
  echo "# tools/probe_train.py --steps 20 (C3 step), default vs HNR_WGRAD_DMA=0";
  for i in 1 2; do python tools/probe_train.py --steps 20 2>/dev/null | tail -1 | cut -c1-200; HNR_WGRAD_DMA=0 python tools/probe_train.py --steps 20 2>/dev/null | tail -1 | cut -c1-200; done; } > $D/r04_wgrad_ab.txt; cat $D/r04_wgrad_ab.txt
