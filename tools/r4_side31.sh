D=gpurun_out/r4_side31d; mkdir -p $D
for i in 1 2; do
  echo "new 15: $(python tools/probe_train.py --steps 40 2>/dev/null | tail -1 | cut -c70-160)"
  echo "new 31: $(HNR_TRAIN_SIDE=31 python tools/probe_train.py --steps 40 2>/dev/null | tail -1 | cut -c70-160)"
  echo "old(pk) 15: $(HNR_LIB_PATH=$PWD/hybridneuralrendering_amd/libhnr_hip_prev.so python tools/probe_train.py --steps 40 2>/dev/null | tail -1 | cut -c70-160)"
done | tee $D/ab.txt
HNR_TRAIN_SIDE=31 RACE_ITERS=10000 timeout 1500 python tools/race_c3.py > $D/race_side31.txt 2>&1; echo "side 31 quiet 10000: $(tail -1 $D/race_side31.txt)"
timeout 900 python tools/stress_determinism.py 2e5 40 render,render > $D/stress_render.txt 2>&1; tail -1 $D/stress_render.txt
(timeout 400 python bench.py --steps 100000 --warmup 1 --no-cpu-baseline --no-train-leg --no-f32-anchor > /dev/null 2>&1 &) ; sleep 30
HNR_TRAIN_SIDE=31 RACE_ITERS=5000 timeout 900 python tools/race_c3.py > $D/race_side31_contended.txt 2>&1; echo "side 31 contended 5000: $(tail -1 $D/race_side31_contended.txt)"
