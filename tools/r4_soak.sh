D=gpurun_out/r4_soak; mkdir -p $D
RACE_ITERS=30000 timeout 1200 python tools/race_c3.py > $D/race_default_quiet.txt 2>&1; echo "default side streams, quiet, 30000 steps: $(tail -1 $D/race_default_quiet.txt)"
timeout 900 python tools/stress_determinism.py 2e5 120 render,render > $D/stress_render.txt 2>&1; tail -1 $D/stress_render.txt
timeout 900 python tools/stress_determinism.py 2e5 120 render,matmul > $D/stress_mixed.txt 2>&1; tail -1 $D/stress_mixed.txt
(timeout 600 python bench.py --steps 100000 --warmup 1 --no-cpu-baseline --no-train-leg --no-f32-anchor > /dev/null 2>&1 &) ; sleep 30
RACE_ITERS=12000 timeout 900 python tools/race_c3.py > $D/race_default_contended.txt 2>&1; echo "default side streams, beside another process, 12000 steps: $(tail -1 $D/race_default_contended.txt)"
