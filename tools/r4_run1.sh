D=gpurun_out/${1:-r4_a}; mkdir -p $D
timeout 900 python -m pytest tests/test_chain_gpu.py -x -q > $D/test_chain.txt 2>&1; tail -5 $D/test_chain.txt
PROBE_MODE=2 timeout 300 python tools/probe_chain.py > $D/probe2.txt 2>&1; tail -5 $D/probe2.txt
timeout 600 python tools/ab_chain.py r3=hybridneuralrendering_amd/libhnr_hip_r3.so new=hybridneuralrendering_amd/libhnr_hip.so --rounds 7 > $D/ab.txt 2>&1; tail -4 $D/ab.txt
