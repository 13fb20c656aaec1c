# round 5: weight gradient with MFMAs and conversion in one phase
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run27; mkdir -p $O
timeout 900 python3 -m pytest tests/test_h2gemm_gpu.py tests/test_train_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -3 $O/pytest.txt
timeout 300 python3 tools/ab_wgrad.py > $O/ab.txt 2>&1; cat $O/ab.txt
HNR_WGRAD_DMA=0 timeout 300 python3 tools/ab_wgrad.py >> $O/ab.txt 2>&1; tail -2 $O/ab.txt
timeout 600 python3 tools/probe_train.py --steps 30 > $O/train.txt 2>&1; tail -2 $O/train.txt
timeout 600 python3 tools/probe_train.py --steps 30 >> $O/train.txt 2>&1; tail -1 $O/train.txt
