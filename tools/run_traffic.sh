# HBM traffic per kernel for the bench frame: separate rocprofv3 --pmc passes (kernel-trace only), each under its own timeout
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic5; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-train-leg --steps 2 --warmup 1 > /tmp/pmc_$c.log 2>&1
  echo "$c rc=$?"
  mkdir -p $OUT/pmc_$c/x; cp /tmp/pmc_$c/*counter_collection.csv $OUT/pmc_$c/x/ 2>/dev/null
done
rm -rf /tmp/st4
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st4 -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-train-leg --steps 4 --warmup 1 > /tmp/st4.log 2>&1
mkdir -p $OUT/stats; cp /tmp/st4/*kernel_stats.csv $OUT/stats/ 2>/dev/null
ls -la $OUT/*/* | head
