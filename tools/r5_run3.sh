# round 5, call 3: neighbourhood-list k-NN (tests + timing vs the 27-cell walk), then the full GPU suite and the training legs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run3; mkdir -p $O
timeout 900 python3 -m pytest tests/test_query_gpu.py -x -q > $O/pytest_query.txt 2>&1; echo "pytest query rc=$?" >> $O/pytest_query.txt; tail -4 $O/pytest_query.txt
for k in 4 3; do for o in 1 0; do
  echo "== HNR_KNN=$k order=$o"; HNR_KNN=$k PROBE_KNN_ORDER=$o timeout 600 python3 tools/probe_query.py 2>&1 | grep -E "grid build|march\+knn" | tee -a $O/query_ab.txt
done; done
timeout 1800 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -8 $O/pytest_gpu.txt
echo "== emu 0/8, 3/8, whole"
for e in 0/8 3/8; do HNR_BENCH_EMULATE_RANK=$e timeout 600 python3 tools/probe_train_shard.py --steps 30 2>>$O/err.txt | tail -1 | tee -a $O/legs.txt; done
timeout 600 python3 tools/probe_train_shard.py --steps 30 2>>$O/err.txt | tail -1 | tee -a $O/legs.txt
HNR_BENCH_TRAIN_GRAPH=0 timeout 600 python3 tools/probe_train.py --steps 20 2>>$O/err.txt | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('ms_per_step','fwd_ms','loss_bwd_ms')})" | tee -a $O/legs.txt
