D=gpurun_out/r4_side63; mkdir -p $D
HNR_TRAIN_SIDE=63 timeout 900 python -m pytest tests/test_train_gpu.py tests/test_sharded_train_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -2
HNR_TRAIN_SIDE=63 RACE_ITERS=8000 timeout 900 python tools/race_c3.py > $D/race63.txt 2>&1; echo "side 63 quiet 8000: $(tail -1 $D/race63.txt)"
(timeout 300 python bench.py --steps 100000 --warmup 1 --no-cpu-baseline --no-train-leg --no-f32-anchor > /dev/null 2>&1 &) ; sleep 30
HNR_TRAIN_SIDE=63 RACE_ITERS=4000 timeout 900 python tools/race_c3.py > $D/race63c.txt 2>&1; echo "side 63 contended 4000: $(tail -1 $D/race63c.txt)"
