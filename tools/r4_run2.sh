D=gpurun_out/${1:-r4_d}; mkdir -p $D
export HNR_LIB_PATH=$PWD/hybridneuralrendering_amd/libhnr_hip_probes.so
for m in 2 3 5 6 7; do echo "== mode $m"; PROBE_MODE=$m timeout 300 python tools/probe_chain.py 2>&1 | tail -5; done > $D/modes.txt 2>&1
echo "== mode 2 TAB0" >> $D/modes.txt; HNR_CHAIN_PROBE_TAB0=1 PROBE_MODE=2 timeout 300 python tools/probe_chain.py 2>&1 | tail -5 >> $D/modes.txt
cat $D/modes.txt
