cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run7; mkdir -p $O
timeout 900 python3 -m pytest tests/test_query_gpu.py tests/test_fullsize_gpu.py -x -q > $O/pytest_query.txt 2>&1; echo "pytest query rc=$?" >> $O/pytest_query.txt; tail -4 $O/pytest_query.txt
for k in 4 6 7; do
  echo "== HNR_KNN=$k order=1"; HNR_KNN=$k PROBE_KNN_ORDER=1 timeout 600 python3 tools/probe_query.py 2>&1 | grep -E "march\+knn" | tee -a $O/query_ab.txt
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk; HNR_KNN=4 PROBE_KNN_ORDER=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_query.py > /tmp/pk.log 2>&1
cp /tmp/pk/*kernel_stats.csv $GRAFT_REPO_ROOT/$O/knn4_kernel_stats.csv; python3 $GRAFT_REPO_ROOT/tools/show_stats.py $GRAFT_REPO_ROOT/$O/knn4_kernel_stats.csv 13 8
