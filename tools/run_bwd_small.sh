cd $GRAFT_REPO_ROOT; OUT=gpurun_out/bwds; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_sharded_train_gpu.py -x -q > $OUT/tests.txt 2>&1; tail -2 $OUT/tests.txt
for m in 0 14 15; do HNR_TRAIN_SIDE=$m timeout 600 python -m pytest tests/test_train_gpu.py -x -q 2>&1 | tail -1; done
for i in 1 2 3; do timeout 300 python tools/probe_train.py --steps 30 2>/dev/null | tail -1 | cut -c90-200; done
RACE_ITERS=1500 timeout 900 python tools/race_c3.py 2>&1 | tail -2
RACE_ITERS=1500 timeout 900 python tools/race_probe.py scannet_small 2>&1 | tail -1
