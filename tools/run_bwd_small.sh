cd $GRAFT_REPO_ROOT; OUT=gpurun_out/bwds; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_sharded_train_gpu.py -x -q > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt
for i in 1 2 3; do timeout 300 python tools/probe_train.py --steps 30 > $OUT/probe_$i.json 2>&1; tail -1 $OUT/probe_$i.json | cut -c1-800; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ct
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ct -o tr -- python3 $GRAFT_REPO_ROOT/tools/probe_train.py --steps 10 > /dev/null 2>/tmp/err_ct.txt
cp /tmp/prof_ct/*kernel_stats.csv $GRAFT_REPO_ROOT/$OUT/kernel_stats.csv
grep -E "ksum|extras" $GRAFT_REPO_ROOT/$OUT/kernel_stats.csv | cut -c1-200
