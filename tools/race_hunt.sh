# which side-stream feature of the training step (HNR_TRAIN_SIDE bit mask) makes the gradient tests flaky
for m in 0 1 2 4 8 15; do
  f=0
  for i in 1 2 3 4 5 6; do
    HNR_TRAIN_SIDE=$m timeout 300 python -m pytest tests/test_train_gpu.py -x -q 2>&1 | grep -q "failed" && f=$((f+1))
  done
  echo "mask $m: $f of 6 runs failed"
done
