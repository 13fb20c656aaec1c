# round 5: spread of the config-3 step over fresh processes on one box (default settings), then the side-stream masks
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run49; mkdir -p $O
for m in 63 63 63 63 63 63 31 31 31; do echo "HNR_TRAIN_SIDE=$m"; HNR_TRAIN_SIDE=$m timeout 600 python3 tools/probe_train.py --steps 30 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['fwd_ms'], d['loss_bwd_ms'])
"; done > $O/train.txt 2>&1
cat $O/train.txt
