cd /tmp && export TMPDIR=/tmp
for k in 10 8 4; do for o in 1; do
echo "== HNR_KNN=$k order=$o pad=0"; PROBE_PAD=0 HNR_KNN=$k PROBE_KNN_ORDER=$o timeout 600 python3 $GRAFT_REPO_ROOT/tools/probe_query.py 2>&1 | grep -E "march\+knn"
done; done
echo "== HNR_KNN=10 order=0 pad=0"; PROBE_PAD=0 HNR_KNN=10 PROBE_KNN_ORDER=0 timeout 600 python3 $GRAFT_REPO_ROOT/tools/probe_query.py 2>&1 | grep -E "march\+knn"
cd $GRAFT_REPO_ROOT; HNR_KNN=10 timeout 900 python3 -m pytest tests/test_query_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -2
HNR_PMC_CMD="$GRAFT_REPO_ROOT/tools/probe_query.py" PROBE_PAD=0 PROBE_KNN_ORDER=1 HNR_KNN=10 bash tools/run_pmc_frame.sh 2>&1 | grep -E "knn_nb|march"
