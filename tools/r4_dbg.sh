HNR_BENCH_REHEARSAL=1 timeout 900 python bench.py --gpus 2 --steps 1 --warmup 0 --points 2e5 --no-cpu-baseline --no-f32-anchor --train-sharded-only 2>&1 | grep -v "^\s*$" | grep DEBUG
