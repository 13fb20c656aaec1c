# round 3 final: the GPU test suite, the default bench line, kernel stats + busy fractions of the frame, emulated 8-way share
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3z
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3z/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r3z/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3z/smoke.txt 2>&1; tail -1 gpurun_out/r3z/smoke.txt
( time python bench.py ) > gpurun_out/r3z/bench_default.json 2> gpurun_out/r3z/bench_default.err; tail -3 gpurun_out/r3z/bench_default.err; cut -c1-400 gpurun_out/r3z/bench_default.json
cd /tmp && export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_f
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -o fr -- python3 $G/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train-leg > $G/gpurun_out/r3z/bench_prof.json 2>/tmp/err_f.txt
cp /tmp/prof_f/*kernel_stats.csv $G/gpurun_out/r3z/bench_kernel_stats.csv 2>/dev/null || tail -3 /tmp/err_f.txt
cd $G && bash tools/run_pmc_frame.sh > gpurun_out/r3z/frame_busy.txt 2>&1; cp gpurun_out/r3f/frame_busy.json gpurun_out/r3z/frame_busy.json; head -12 gpurun_out/r3z/frame_busy.txt
bash tools/predict_scaling.sh > gpurun_out/r3z/predict.log 2>&1; tail -8 gpurun_out/r3z/predict.log | cut -c1-60
