cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run12; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_train_gpu.py tests/test_query_gpu.py tests/test_h2gemm_gpu.py tests/test_bench_gpu.py -x -q -k "exchange or variant or dma_staged or rehearsal_of_the_patch" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -12 $O/pytest.txt
