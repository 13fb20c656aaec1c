cd $GRAFT_REPO_ROOT
bash tools/run_trace_train.sh > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ct
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ct -o tr -- python3 $GRAFT_REPO_ROOT/tools/probe_train.py --steps 10 > /dev/null 2>/tmp/err_ct.txt
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r3z; cp /tmp/prof_ct/*kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/r3z/train_kernel_stats.csv
cd $GRAFT_REPO_ROOT; for i in 1 2 3; do timeout 300 python tools/probe_train.py --steps 30 2>/dev/null | tail -1 | cut -c90-200; done
