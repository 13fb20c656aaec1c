# VALU wave-instructions of the k-NN kernel per launch, reference slot order vs sorted order (rocprofv3 --pmc, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_knn; mkdir -p $OUT
for o in 0 1; do
  rm -rf /tmp/pk$o
  PROBE_KNN_ORDER=$o timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d /tmp/pk$o -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_query.py > /tmp/pk$o.log 2>&1
  cp /tmp/pk$o/*counter_collection.csv $OUT/order$o.csv 2>/dev/null || tail -5 /tmp/pk$o.log
done
python3 - <<'PY'
import csv, collections, os
root = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_knn")
for o in (0, 1):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(os.path.join(root, "order%d.csv" % o))):
        if "knn" in r["Kernel_Name"] or "march" in r["Kernel_Name"]:
            per[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        print("order %d %-40s %s" % (o, k[:40], {c: "%.1f M" % (sum(x) / len(x) / 1e6) for c, x in v.items()}))
PY
