D=gpurun_out/r4_featmap; mkdir -p $D
echo "shipped build (packed fp32 in featmap_kernel)"; FM_ITERS=2000 python tools/featmap_contention.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee $D/featmap_pk.txt
echo "aggregate.hip with -fno-slp-vectorize (no v_pk_*_f32)"; HNR_LIB_PATH=$PWD/hybridneuralrendering_amd/libhnr_hip_noslp.so FM_ITERS=2000 python tools/featmap_contention.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee $D/featmap_noslp.txt
