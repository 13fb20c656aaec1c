# round 5: HBM traffic of the training step's large kernels after the chain moved (profiles/r05_train_traffic.json), kernel stats of the training step
cd $GRAFT_REPO_ROOT
bash tools/run_train_traffic.sh r05 > gpurun_out/r5_run39_traffic.txt 2>&1; tail -14 gpurun_out/r5_run39_traffic.txt | cut -c1-150
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/st6
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st6 -o s -- python3 $GRAFT_REPO_ROOT/tools/probe_train.py --steps 10 > /tmp/st6.log 2>&1
cp /tmp/st6/*kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/r5_run39_train_kernel_stats.csv 2>/dev/null; head -12 $GRAFT_REPO_ROOT/gpurun_out/r5_run39_train_kernel_stats.csv | cut -c1-140
cp $GRAFT_REPO_ROOT/profiles/r05_train_traffic.json $GRAFT_REPO_ROOT/gpurun_out/ 2>/dev/null
