D=gpurun_out/${1:-r4_stg}; mkdir -p $D
A=hybridneuralrendering_amd
timeout 900 python tools/ab_chain.py base=$A/libhnr_hip_prev.so stg1=$A/libhnr_hip_stg1.so stg2=$A/libhnr_hip_stg2.so stg4=$A/libhnr_hip_stg4.so --rounds 9 > $D/ab.txt 2>&1; tail -7 $D/ab.txt
