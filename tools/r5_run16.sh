cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run16; mkdir -p $O; rm -f $O/legs.txt
timeout 1200 python3 -m pytest tests/test_train_gpu.py tests/test_fullsize_gpu.py tests/test_sharded_train_gpu.py -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
for e in 0/8 3/8; do HNR_BENCH_EMULATE_RANK=$e timeout 600 python3 tools/probe_train_shard.py --steps 30 2>>$O/err.txt | tail -1 | tee -a $O/legs.txt; done
timeout 600 python3 tools/probe_train_shard.py --steps 30 2>>$O/err.txt | tail -1 | tee -a $O/legs.txt
HNR_BENCH_TRAIN_GRAPH=0 timeout 600 python3 tools/probe_train.py --steps 20 2>>$O/err.txt | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('ms_per_step','fwd_ms','loss_bwd_ms')})" | tee -a $O/legs.txt
