cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run17; mkdir -p $O
rm -rf /tmp/prof_c5
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c5 -o tr -- python3 $GRAFT_REPO_ROOT/tools/probe_train_shard.py --steps 4 > /tmp/c5.json 2>/tmp/c5.err
cd $GRAFT_REPO_ROOT
f=$(ls /tmp/prof_c5/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python3 tools/timeline.py $f $O/c5_timeline.txt march_kernel 14 > /dev/null; tail -1 $O/c5_timeline.txt
