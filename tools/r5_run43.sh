# round 5: hnr_h2lin with the narrow layers' (row tile, column tile) pairs dealt to all four waves: tests, same-box A/B (HNR_H2LIN_SPREAD)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run43; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_h2gemm_gpu.py tests/test_train_gpu.py tests/test_sharded_train_gpu.py tests/test_fullsize_gpu.py tests/test_linear_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -4 $O/pytest.txt
for v in 1 0 1 0; do echo "HNR_H2LIN_SPREAD=$v"; HNR_H2LIN_SPREAD=$v timeout 600 python3 tools/probe_train.py --steps 30 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['loss_bwd_ms'], {k: round(v, 3) for k, v in d['stage_ms'].items() if k.startswith('bwd.')})
"; done > $O/train.txt 2>&1
cat $O/train.txt
