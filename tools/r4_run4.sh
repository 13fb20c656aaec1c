D=gpurun_out/${1:-r4_h}; mkdir -p $D
timeout 2400 python -m pytest tests/test_bench_gpu.py tests/test_train_gpu.py tests/test_query_gpu.py -x -q > $D/pytest.txt 2>&1; tail -6 $D/pytest.txt
bash tools/predict_train_scaling.sh 8 > $D/predict_train_8.txt 2>&1; cat $D/predict_train_8.txt | cut -c1-400
