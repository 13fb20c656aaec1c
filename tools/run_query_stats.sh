# per-kernel durations of the query stage (rocprofv3 --kernel-trace --stats), reference slot order vs sorted order
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/query_stats; mkdir -p $OUT
for o in 0 1; do
  rm -rf /tmp/qs$o
  PROBE_KNN_ORDER=$o timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qs$o -o q -- python3 $GRAFT_REPO_ROOT/tools/probe_query.py > /tmp/qs$o.log 2>&1
  cp /tmp/qs$o/*kernel_stats.csv $OUT/order$o.csv 2>/dev/null || tail -5 /tmp/qs$o.log
  grep -h "knn3\|march_kernel\|worklist" $OUT/order$o.csv | awk -F'","' '{print substr($1,1,50), $2, $4}'
done
