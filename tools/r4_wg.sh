timeout 900 python -m pytest tests/test_h2gemm_gpu.py -x -q 2>&1 | tail -3
python tools/ab_wgrad.py 2>&1 | grep HNR_; HNR_WGRAD_DMA=0 python tools/ab_wgrad.py 2>&1 | grep HNR_
python tools/ab_wgrad.py 2>&1 | grep HNR_; HNR_WGRAD_DMA=0 python tools/ab_wgrad.py 2>&1 | grep HNR_
for i in 1 2; do python tools/probe_train.py --steps 20 2>/dev/null | tail -1 | cut -c1-200; HNR_WGRAD_DMA=0 python tools/probe_train.py --steps 20 2>/dev/null | tail -1 | cut -c1-200; done
