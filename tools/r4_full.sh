D=gpurun_out/${1:-r4_full}; mkdir -p $D
timeout 2400 python -m pytest tests -m gpu -x -q > $D/pytest_gpu.txt 2>&1; tail -5 $D/pytest_gpu.txt
timeout 900 python bench.py > $D/bench.json 2> $D/bench.err; tail -c 3000 $D/bench.json
