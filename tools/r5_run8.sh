cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run8; mkdir -p $O
for k in 4 3; do
echo "== HNR_KNN=$k pad=0"; PROBE_PAD=0 HNR_KNN=$k PROBE_KNN_ORDER=1 timeout 600 python3 $GRAFT_REPO_ROOT/tools/probe_query.py 2>&1 | grep -E "march\+knn"
done
rm -rf /tmp/pk0; PROBE_PAD=0 HNR_KNN=4 PROBE_KNN_ORDER=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk0 -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_query.py > /tmp/pk0.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/show_stats.py /tmp/pk0/*kernel_stats.csv 13 12 | grep -E "march|knn_quad|worklist|nsamp|fill"
cp /tmp/pk0/*kernel_stats.csv $O/query_nopad_kernel_stats.csv
