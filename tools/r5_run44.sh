cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run44; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python3 $GRAFT_REPO_ROOT/tools/probe_train.py --steps 6 > /tmp/tl.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/timeline.py $(ls /tmp/tl/*kernel_trace.csv | head -1) $GRAFT_REPO_ROOT/$O/timeline.txt > /dev/null 2>&1; tail -2 $GRAFT_REPO_ROOT/$O/timeline.txt
