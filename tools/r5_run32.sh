cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run32; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_train_gpu.py tests/test_sharded_train_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -5 $O/pytest.txt
