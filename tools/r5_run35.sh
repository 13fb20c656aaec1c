cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run35; mkdir -p $O
for i in 1 2; do timeout 600 python3 tools/probe_train.py --steps 30 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['fwd_ms'], d['loss_bwd_ms'], {k: round(v, 3) for k, v in d['stage_ms'].items()})
"; done > $O/train.txt 2>&1
cat $O/train.txt
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python3 $GRAFT_REPO_ROOT/tools/probe_train.py --steps 6 > /tmp/tl.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/timeline.py $(ls /tmp/tl/*kernel_trace.csv | head -1) $GRAFT_REPO_ROOT/$O/timeline.txt > /dev/null 2>&1; tail -3 $GRAFT_REPO_ROOT/$O/timeline.txt
