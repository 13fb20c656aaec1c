cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run13; mkdir -p $O
rm -rf /tmp/prof_c3
HNR_BENCH_TRAIN_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c3 -o tr -- python3 $GRAFT_REPO_ROOT/tools/probe_train.py --steps 4 > /tmp/c3.json 2>/tmp/c3.err
cd $GRAFT_REPO_ROOT
f=$(ls /tmp/prof_c3/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python3 tools/timeline.py $f $O/c3_timeline.txt march_kernel 14 > /dev/null; tail -1 $O/c3_timeline.txt
