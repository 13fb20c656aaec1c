# train-step kernel stats of the two-call training path + the whole GPU suite
cd /tmp && export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT; OUT=$G/gpurun_out/r3b; mkdir -p $OUT
cd $G && timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/tests.log; cat $OUT/tests.log | tail -8
cd /tmp; rm -rf /tmp/prof_t
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o tr -- python3 $G/tools/probe_train.py --steps 10 > $OUT/probe_train.json 2>/tmp/err_t.txt
cp /tmp/prof_t/*kernel_stats.csv $OUT/train_v1_kernel_stats.csv 2>/dev/null; tail -2 /tmp/err_t.txt; cat $OUT/probe_train.json | cut -c1-400
timeout 300 python3 $G/tools/probe_train.py --steps 20 > $OUT/probe_train_noprof.json 2>/tmp/err_t2.txt; cat $OUT/probe_train_noprof.json | cut -c1-300
