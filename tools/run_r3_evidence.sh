# round 3 evidence: HBM traffic (PMC, separate passes), kernel stats of the bench frame, power / clock trace, emulated 8-way share
cd /tmp && export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT
bash $G/tools/run_traffic.sh > $G/gpurun_out/traffic5.log 2>&1; tail -3 $G/gpurun_out/traffic5.log
mkdir -p $G/gpurun_out/r3e
timeout 600 python3 $G/tools/power_trace.py --steps 300 --out $G/gpurun_out/r3e/power_trace.csv > $G/gpurun_out/r3e/power_summary.json 2>$G/gpurun_out/r3e/power_err.txt
cat $G/gpurun_out/r3e/power_summary.json | cut -c1-600
cd $G && bash tools/predict_scaling.sh > gpurun_out/r3e/predict.log 2>&1; tail -12 gpurun_out/r3e/predict.log
