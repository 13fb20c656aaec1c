cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run34; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_train_gpu.py tests/test_sharded_train_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -5 $O/pytest.txt
for v in 1 0 1; do echo "HNR_TRAIN_CHAIN_WS=$v"; HNR_TRAIN_CHAIN_WS=$v timeout 600 python3 tools/probe_train.py --steps 30 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['fwd_ms'], d['stage_ms'].get('fwd.chain'))
"; done > $O/train.txt 2>&1
cat $O/train.txt
