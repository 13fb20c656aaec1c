D=gpurun_out/r4_side31; mkdir -p $D
for i in 1 2; do
  python tools/probe_train.py --steps 40 2>/dev/null | tail -1 | cut -c1-160
  HNR_TRAIN_SIDE=31 python tools/probe_train.py --steps 40 2>/dev/null | tail -1 | cut -c1-160
done > $D/ab.txt; cat $D/ab.txt
HNR_TRAIN_SIDE=31 RACE_ITERS=15000 timeout 1500 python tools/race_c3.py > $D/race_side31.txt 2>&1; tail -2 $D/race_side31.txt
# the same three queues while ANOTHER process renders on the GPU (the contention that showed the division anomaly)
(timeout 400 python bench.py --steps 8000 --warmup 1 --no-cpu-baseline --no-train-leg --no-f32-anchor > /dev/null 2>&1 &) ; sleep 25
HNR_TRAIN_SIDE=31 RACE_ITERS=4000 timeout 900 python tools/race_c3.py > $D/race_side31_contended.txt 2>&1; tail -2 $D/race_side31_contended.txt
