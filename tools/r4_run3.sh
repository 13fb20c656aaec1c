D=gpurun_out/${1:-r4_e}; mkdir -p $D
export HNR_LIB_PATH=$PWD/hybridneuralrendering_amd/libhnr_hip_probes.so
for m in 4; do echo "== mode $m"; PROBE_MODE=$m timeout 300 python tools/probe_chain.py 2>&1 | tail -5; done > $D/modes4.txt 2>&1
cat $D/modes4.txt
