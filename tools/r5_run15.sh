cd /tmp && export TMPDIR=/tmp
for k in 4 8; do
echo "== HNR_KNN=$k order=1 pad=0"; PROBE_PAD=0 HNR_KNN=$k PROBE_KNN_ORDER=1 timeout 600 python3 $GRAFT_REPO_ROOT/tools/probe_query.py 2>&1 | grep -E "march\+knn"
done
echo "== HNR_KNN=4 wg 7"; HNR_KNN_WG_PER_CU=7 PROBE_PAD=0 HNR_KNN=4 PROBE_KNN_ORDER=1 timeout 600 python3 $GRAFT_REPO_ROOT/tools/probe_query.py 2>&1 | grep -E "march\+knn"
cd $GRAFT_REPO_ROOT; timeout 900 python3 -m pytest tests/test_query_gpu.py -x -q 2>&1 | tail -2
