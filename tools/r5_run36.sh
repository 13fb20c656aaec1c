cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run36; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_train_gpu.py tests/test_sharded_train_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -4 $O/pytest.txt
for i in 1 2; do timeout 600 python3 tools/probe_train.py --steps 30 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['fwd_ms'], d['loss_bwd_ms'], {k: round(v, 3) for k, v in d['stage_ms'].items()})
"; done > $O/train.txt 2>&1
cat $O/train.txt
timeout 900 python3 tools/probe_train_shard.py > $O/shard.txt 2>&1; tail -5 $O/shard.txt
