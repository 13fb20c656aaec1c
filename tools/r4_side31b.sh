D=gpurun_out/r4_side31c; mkdir -p $D
(timeout 300 python bench.py --steps 12000 --warmup 1 --no-cpu-baseline --no-train-leg --no-f32-anchor > /dev/null 2>&1 &) ; sleep 25
RACE_VERBOSE=1 HNR_TRAIN_SIDE=0 RACE_ITERS=1200 timeout 600 python tools/race_c3.py > $D/c0.txt 2>&1; tail -1 $D/c0.txt
grep "^step" $D/c0.txt | head -5 | python -c "
import sys,ast
for l in sys.stdin:
    i=l.index('['); lst=ast.literal_eval(l[i:]); print(l[:i], [x for x in lst if x[0].startswith('out.') or x[0]=='coarse_raycolor'])"
