# round 5: ablation of the DMA weight-gradient kernel (probe build -DHNR_WG_DBG: results are garbage with a bit set)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run28; mkdir -p $O
for d in 0 1 2 4 3 5 6 7; do echo "== HNR_WG_DBG=$d (1 no MFMAs, 2 no conversion, 4 no loads in the loop)"; HNR_WG_DBG=$d timeout 300 python3 tools/ab_wgrad.py 2>&1 | grep "K=256"; done > $O/abl.txt 2>&1
cat $O/abl.txt
