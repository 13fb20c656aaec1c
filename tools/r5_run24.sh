cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run24; mkdir -p $O; rm -f $O/legs.txt
for rt in 4 2; do
echo "== HNR_TRAIN_CHAIN_RT=$rt"
HNR_TRAIN_CHAIN_RT=$rt HNR_BENCH_TRAIN_GRAPH=0 timeout 600 python3 tools/probe_train.py --steps 20 2>>$O/err.txt | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('ms_per_step','fwd_ms','loss_bwd_ms')}, {k:v for k,v in d['stage_ms'].items() if k.startswith('fwd.')})" | tee -a $O/legs.txt
done
HNR_TRAIN_CHAIN_RT=2 timeout 1200 python3 -m pytest tests/test_train_gpu.py -x -q 2>&1 | tail -3
