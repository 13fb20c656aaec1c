cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run31; mkdir -p $O
for v in 1 0 1; do echo "HNR_TRAIN_CHAIN_WS=$v"; HNR_TRAIN_CHAIN_WS=$v timeout 600 python3 tools/probe_train.py --steps 30 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['fwd_ms'], d['stage_ms'].get('fwd.chain'))
"; done > $O/train.txt 2>&1
cat $O/train.txt
