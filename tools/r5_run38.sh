# round 5 (after the training chain moved to chain_ws_kernel<8> + sign words): emulated 1/8 shares of the C5 step, the whole C5 batch, C3
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_run38; mkdir -p $O
for r in 0 3 7; do HNR_BENCH_EMULATE_RANK=$r/8 timeout 600 python3 tools/probe_train_shard.py --steps 30 2>/dev/null | grep "^{" ; done > $O/predict.txt
timeout 600 python3 tools/probe_train_shard.py --steps 30 2>/dev/null | grep "^{" >> $O/predict.txt
timeout 600 python3 tools/probe_train.py --steps 30 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print({k: d[k] for k in ('ms_per_step', 'fwd_ms', 'loss_bwd_ms')})
" >> $O/predict.txt
cat $O/predict.txt | cut -c1-260
