cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run10; mkdir -p $O
rm -rf /tmp/pk0; PROBE_PAD=0 HNR_KNN=4 PROBE_KNN_ORDER=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk0 -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_query.py > /tmp/pk0.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/show_stats.py /tmp/pk0/*kernel_stats.csv 13 12 | grep -E "march|knn_quad|worklist|nsamp|fill"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pq$i
  PROBE_PAD=0 HNR_KNN=4 PROBE_KNN_ORDER=1 timeout 900 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pq$i -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_query.py > /tmp/pq$i.log 2>&1
  cp /tmp/pq$i/*counter_collection.csv $O/pmc_g$i.csv 2>/dev/null || tail -5 /tmp/pq$i.log
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r5_run10/pmc_g*.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        if "knn" in k or "march" in k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in acc:
        print(f.split("/")[-1], k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
