D=gpurun_out/r4_zero; mkdir -p $D
timeout 1200 python -m pytest tests/test_modules_gpu.py tests/test_render_gpu.py tests/test_mlp_gpu.py -x -q 2>&1 | tail -2
for i in 1 2 3; do
for lib in prev new; do
  L=$PWD/hybridneuralrendering_amd/libhnr_hip.so; [ $lib = prev ] && L=$PWD/hybridneuralrendering_amd/libhnr_hip_prev.so
  HNR_LIB_PATH=$L python bench.py --no-cpu-baseline --no-f32-anchor --no-train-leg --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.strip().startswith('{')][0]); s=d['stage_ms']; print('$lib', round(d['ms_per_step'],3), {k:round(s[k],3) for k in ('chain','mlp_colorfeat','mlp_merge','mlp_mixup')})"
done; done | tee $D/ab.txt
