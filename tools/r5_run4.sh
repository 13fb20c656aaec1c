# round 5, call 4: the neighbourhood-list k-NN with the in-block sort by list length (HNR_KNN=4) vs work-list order (5) vs the 27-cell walk (3); PMC of (4)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run4; mkdir -p $O
timeout 900 python3 -m pytest tests/test_query_gpu.py tests/test_fullsize_gpu.py -x -q > $O/pytest_query.txt 2>&1; echo "pytest query rc=$?" >> $O/pytest_query.txt; tail -4 $O/pytest_query.txt
for k in 4 5 3; do
  echo "== HNR_KNN=$k order=1"; HNR_KNN=$k PROBE_KNN_ORDER=1 timeout 600 python3 tools/probe_query.py 2>&1 | grep -E "march\+knn" | tee -a $O/query_ab.txt
done
echo "== HNR_KNN=4 order=0"; HNR_KNN=4 PROBE_KNN_ORDER=0 timeout 600 python3 tools/probe_query.py 2>&1 | grep -E "march\+knn" | tee -a $O/query_ab.txt
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1)); rm -rf /tmp/pq$i
  PROBE_KNN_ORDER=1 timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pq$i -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_query.py > /tmp/pq$i.log 2>&1
  cp /tmp/pq$i/*counter_collection.csv $GRAFT_REPO_ROOT/$O/pmc_g$i.csv 2>/dev/null || tail -5 /tmp/pq$i.log
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r5_run4/pmc_g*.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        if "knn" in k or "march" in k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in acc:
        print(f.split("/")[-1], k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
