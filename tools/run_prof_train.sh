cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_t
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o tr -- python3 $GRAFT_REPO_ROOT/tools/probe_train.py --steps 10 > $GRAFT_REPO_ROOT/gpurun_out/probe_train.json 2>/tmp/err.txt
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_t; cp /tmp/prof_t/*kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/prof_t/
cat $GRAFT_REPO_ROOT/gpurun_out/probe_train.json; tail -3 /tmp/err.txt
