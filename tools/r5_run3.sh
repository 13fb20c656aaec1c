# round 5, call 3: full GPU suite on the graph-free step + exchange kernels + fast blur_select; training legs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run3; mkdir -p $O
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -8 $O/pytest_gpu.txt
echo "== emu 0/8, 3/8, whole"
for e in 0/8 3/8; do HNR_BENCH_EMULATE_RANK=$e timeout 600 python3 tools/probe_train_shard.py --steps 30 2>>$O/err.txt | tail -1 | tee -a $O/legs.txt; done
timeout 600 python3 tools/probe_train_shard.py --steps 30 2>>$O/err.txt | tail -1 | tee -a $O/legs.txt
HNR_BENCH_TRAIN_GRAPH=0 timeout 600 python3 tools/probe_train.py --steps 20 2>>$O/err.txt | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('ms_per_step','fwd_ms','loss_bwd_ms')})" | tee -a $O/legs.txt
