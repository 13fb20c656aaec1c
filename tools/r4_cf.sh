timeout 600 python tools/ab_cf.py 200000 2>&1 | grep HNR_; HNR_CF_WS=0 timeout 600 python tools/ab_cf.py 200000 2>&1 | grep HNR_
timeout 600 python tools/ab_cf.py 2>&1 | grep HNR_; HNR_CF_WS=0 timeout 600 python tools/ab_cf.py 2>&1 | grep HNR_
HNR_LIB_PATH=$PWD/hybridneuralrendering_amd/libhnr_hip_cfprobe.so python tools/ab_cf.py 2>&1 | grep "probe" | tail -1
