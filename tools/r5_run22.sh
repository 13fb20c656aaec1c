cd /tmp && export TMPDIR=/tmp
for m in 0 4; do
echo "== HNR_MARCH_PROBE=$m pad=0"; HNR_MARCH_PROBE=$m PROBE_PAD=0 PROBE_KNN_ORDER=1 timeout 600 python3 $GRAFT_REPO_ROOT/tools/probe_query.py 2>&1 | grep -E "march\+knn|grid build"
done
cd $GRAFT_REPO_ROOT; timeout 900 python3 -m pytest tests/test_query_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -2
cd /tmp; rm -rf /tmp/pk0; PROBE_PAD=0 PROBE_KNN_ORDER=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk0 -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_query.py > /tmp/pk0.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/show_stats.py /tmp/pk0/*kernel_stats.csv 13 14 | grep -E "march|knn_|worklist|brick_near"
