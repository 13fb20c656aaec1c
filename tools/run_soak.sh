# run-to-run bit checks of the training step (side-stream configuration from HNR_TRAIN_SIDE), bench batch and the golden batches
cd $GRAFT_REPO_ROOT
rocm-smi --showserial 2>/dev/null | grep -i serial | head -1
for r in 1 2; do
RACE_ITERS=3000 timeout 900 python tools/race_c3.py 2>&1 | tail -2
RACE_ITERS=3000 timeout 900 python tools/race_probe.py scannet_small 2>&1 | tail -2
RACE_ITERS=3000 timeout 900 python tools/race_probe.py 2>&1 | tail -2
RACE_ITERS=2000 timeout 900 python tools/race_probe.py synth_small 2>&1 | tail -2
done
