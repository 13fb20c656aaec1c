D=gpurun_out/r4_split; mkdir -p $D
timeout 2400 python -m pytest tests -m gpu -x -q > $D/pytest_gpu.txt 2>&1; tail -3 $D/pytest_gpu.txt
for i in 1 2; do
for lib in prev new; do
  L=$PWD/hybridneuralrendering_amd/libhnr_hip.so; [ $lib = prev ] && L=$PWD/hybridneuralrendering_amd/libhnr_hip_prev.so
  HNR_LIB_PATH=$L python bench.py --no-cpu-baseline --no-f32-anchor --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.strip().startswith('{')][0]); s=d['stage_ms']; print('$lib', round(d['ms_per_step'],3), {k:round(s[k],3) for k in ('chain_gather','chain','mlp_colorfeat','mlp_merge','mlp_mixup')}, 'train', d['train_step']['ms_per_step'], d['train_step']['fwd_ms'])"
done; done | tee $D/ab.txt
