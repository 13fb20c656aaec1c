D=gpurun_out/${1:-r4_div}; mkdir -p $D
timeout 900 python -m pytest tests/test_render_gpu.py tests/test_chain_gpu.py tests/test_modules_gpu.py -x -q 2>&1 | tail -3
# shared-GPU renders: the r2 experiment that showed 7-14 of 40 renders differing (chain_gather's perspective division, lanes 48..63)
timeout 900 python tools/stress_determinism.py 2e5 40 render,render > $D/stress_render.txt 2>&1; tail -3 $D/stress_render.txt
timeout 900 python tools/stress_determinism.py 2e5 40 render,matmul > $D/stress_mixed.txt 2>&1; tail -2 $D/stress_mixed.txt
# three busy queues in the training step (HNR_TRAIN_SIDE=31) + default
HNR_TRAIN_SIDE=31 RACE_ITERS=3000 timeout 900 python tools/race_c3.py > $D/race_c3_side31.txt 2>&1; tail -2 $D/race_c3_side31.txt
RACE_ITERS=2000 timeout 900 python tools/race_c3.py > $D/race_c3_default.txt 2>&1; tail -2 $D/race_c3_default.txt
