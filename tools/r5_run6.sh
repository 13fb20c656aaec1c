cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run6; mkdir -p $O
for k in 4 7; do
rm -rf /tmp/pk$k
HNR_KNN=$k PROBE_KNN_ORDER=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk$k -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_query.py > /tmp/pk$k.log 2>&1
cp /tmp/pk$k/*kernel_stats.csv $O/knn${k}_kernel_stats.csv
python3 $GRAFT_REPO_ROOT/tools/show_stats.py $O/knn${k}_kernel_stats.csv 13 14
done
