#!/bin/bash
# ablation of the merge stage kernel: which of gather / weight stream / stores the 2.7 ms are
for a in 0 1 2 3 4 7; do
  HNR_MW_ABL=$a python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-train-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('abl $a', d['ms_per_step'], d['stage_ms']['mlp_merge'])"
done
