# HBM traffic per kernel of the training step: separate rocprofv3 --pmc passes (kernel-trace only) over tools/probe_train.py
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/train_traffic; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmct_$c
  timeout 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmct_$c -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_train.py --steps 4 > /tmp/pmct_$c.log 2>&1
  echo "$c rc=$?"
  mkdir -p $OUT/pmc_$c/x; cp /tmp/pmct_$c/*counter_collection.csv $OUT/pmc_$c/x/ 2>/dev/null
done
cd $GRAFT_REPO_ROOT && python tools/collect_train_traffic.py gpurun_out/train_traffic ${1:-r05}
