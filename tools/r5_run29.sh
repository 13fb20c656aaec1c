# round 5: DMA weight gradient, loads three blocks ahead (HNR_WGRAD_DMA=2) vs one phase (1) vs register-staged (0)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run29; mkdir -p $O
for d in 3 2 1 3; do HNR_WGRAD_DMA=$d timeout 300 python3 tools/ab_wgrad.py 2>&1 | grep "K=2"; done > $O/ab.txt 2>&1
cat $O/ab.txt
for d in 3 1; do echo "HNR_WGRAD_DMA=$d"; HNR_WGRAD_DMA=$d timeout 600 python3 tools/probe_train.py --steps 30 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline_train']['avg_launch_ms'], d['roofline_train']['frac'])
"; done > $O/train.txt 2>&1
cat $O/train.txt
