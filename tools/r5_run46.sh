# round 5: the training tests on the fallback chain kernel (HNR_TRAIN_CHAIN_WS=0: chain_kernel<4,3> + sign words) and without side streams
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run46; mkdir -p $O
HNR_TRAIN_CHAIN_WS=0 timeout 1500 python3 -m pytest tests/test_train_gpu.py tests/test_sharded_train_gpu.py tests/test_fullsize_gpu.py tests/test_modules_gpu.py -x -q -m gpu > $O/pytest_ws0.txt 2>&1; echo "ws0 rc=$?"; tail -2 $O/pytest_ws0.txt
HNR_TRAIN_SIDE=0 timeout 1500 python3 -m pytest tests/test_train_gpu.py tests/test_sharded_train_gpu.py -x -q -m gpu > $O/pytest_side0.txt 2>&1; echo "side0 rc=$?"; tail -2 $O/pytest_side0.txt
