# round 5, call 2: where does a 1/8 share of the C5 step spend its time?  graph vs eager x side streams on / off, and a kernel timeline of the eager share
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run2; mkdir -p $O
timeout 900 python3 -m pytest tests/test_train_gpu.py -x -q -k "blur_module or captured" > $O/pytest_new.txt 2>&1; echo "pytest new rc=$?" >> $O/pytest_new.txt
tail -5 $O/pytest_new.txt
for side in 63 0; do for gr in 1 0; do
  echo "== emu 0/8 HNR_TRAIN_SIDE=$side graph=$gr"
  HNR_TRAIN_SIDE=$side HNR_BENCH_TRAIN_GRAPH=$gr HNR_BENCH_EMULATE_RANK=0/8 timeout 600 python3 tools/probe_train_shard.py --steps 30 2>>$O/matrix.err | tail -1 | tee -a $O/matrix.txt
done; done
echo "== whole C5 batch eager / graph"
HNR_BENCH_TRAIN_GRAPH=0 timeout 600 python3 tools/probe_train_shard.py --steps 30 2>>$O/matrix.err | tail -1 | tee -a $O/matrix.txt
HNR_BENCH_TRAIN_GRAPH=1 timeout 600 python3 tools/probe_train_shard.py --steps 30 2>>$O/matrix.err | tail -1 | tee -a $O/matrix.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ts
HNR_BENCH_TRAIN_GRAPH=0 HNR_BENCH_EMULATE_RANK=0/8 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ts -o tr -- python3 $GRAFT_REPO_ROOT/tools/probe_train_shard.py --steps 4 > /tmp/ts.json 2>/tmp/ts.err
cd $GRAFT_REPO_ROOT
f=$(ls /tmp/prof_ts/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python3 tools/timeline.py $f $O/emu0_eager_timeline.txt march_kernel 12 | tail -150
