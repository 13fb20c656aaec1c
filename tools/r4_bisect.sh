D=gpurun_out/r4_bisect; mkdir -p $D
RACE_ITERS=20 timeout 300 python tools/bisect_train_forward.py 2>&1 | grep -v amdgpu.ids | tail -3
(timeout 300 python bench.py --steps 12000 --warmup 1 --no-cpu-baseline --no-train-leg --no-f32-anchor > /dev/null 2>&1 &) ; sleep 25
HNR_TRAIN_SIDE=0 RACE_ITERS=800 timeout 600 python tools/bisect_train_forward.py > $D/side0.txt 2>&1; grep -v amdgpu.ids $D/side0.txt | tail -17 | cut -c1-700
