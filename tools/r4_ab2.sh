D=gpurun_out/${1:-r4_ab5}; mkdir -p $D
timeout 600 python -m pytest tests/test_chain_gpu.py -x -q 2>&1 | tail -2
PROBE_MODE=2 timeout 300 python tools/probe_chain.py > $D/probe2.txt 2>&1; tail -5 $D/probe2.txt
timeout 900 python tools/ab_chain.py prev=hybridneuralrendering_amd/libhnr_hip_prev.so new=hybridneuralrendering_amd/libhnr_hip.so --rounds 9 > $D/ab.txt 2>&1; tail -4 $D/ab.txt
