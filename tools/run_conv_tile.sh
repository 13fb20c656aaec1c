# tiled conv backward: unit test (tiled and per-element forms), training gradient tests, step time both ways, kernel stats
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/convt; mkdir -p $OUT
timeout 600 python -m pytest tests/test_conv_bwd_gpu.py -x -q > $OUT/unit_tiled.txt 2>&1; tail -5 $OUT/unit_tiled.txt
HNR_CONV_BWD_NAIVE=1 timeout 600 python -m pytest tests/test_conv_bwd_gpu.py -x -q > $OUT/unit_naive.txt 2>&1; tail -3 $OUT/unit_naive.txt
timeout 900 python -m pytest tests/test_train_gpu.py -x -q > $OUT/train_tests.txt 2>&1; tail -5 $OUT/train_tests.txt
for i in 1 2; do
timeout 300 python tools/probe_train.py --steps 30 > $OUT/probe_tiled_$i.json 2>&1; tail -1 $OUT/probe_tiled_$i.json | cut -c1-400
HNR_CONV_BWD_NAIVE=1 timeout 300 python tools/probe_train.py --steps 30 > $OUT/probe_naive_$i.json 2>&1; tail -1 $OUT/probe_naive_$i.json | cut -c1-400
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ct
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ct -o tr -- python3 $GRAFT_REPO_ROOT/tools/probe_train.py --steps 10 > /dev/null 2>/tmp/err_ct.txt
cp /tmp/prof_ct/*kernel_stats.csv $GRAFT_REPO_ROOT/$OUT/kernel_stats.csv
head -40 $GRAFT_REPO_ROOT/$OUT/kernel_stats.csv | cut -c1-160
