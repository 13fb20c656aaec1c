cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run5; mkdir -p $O
timeout 900 python3 -m pytest tests/test_query_gpu.py -x -q > $O/pytest_query.txt 2>&1; echo "pytest query rc=$?" >> $O/pytest_query.txt; tail -4 $O/pytest_query.txt
for k in 4 6 7 5 3; do
  echo "== HNR_KNN=$k order=1"; HNR_KNN=$k PROBE_KNN_ORDER=1 timeout 600 python3 tools/probe_query.py 2>&1 | grep -E "march\+knn" | tee -a $O/query_ab.txt
done
