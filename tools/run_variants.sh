# training tests under the side-stream switches, and the new direct tests
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/variants; mkdir -p $OUT
timeout 600 python -m pytest tests/test_merge_bwd_gpu.py tests/test_composite_bwd_gpu.py tests/test_conv_bwd_gpu.py -x -q > $OUT/direct.txt 2>&1; tail -12 $OUT/direct.txt
for m in 0 1 2 8 3; do HNR_TRAIN_SIDE=$m timeout 900 python -m pytest tests/test_train_gpu.py tests/test_sharded_train_gpu.py -x -q > $OUT/side_$m.txt 2>&1; echo "HNR_TRAIN_SIDE=$m: $(tail -1 $OUT/side_$m.txt)"; done
