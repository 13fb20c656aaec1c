#!/bin/bash
# Ablation timings of the split-bf16 dense layer (probe build: csrc/build/libhnr_probe.so, -DHNR_LINEAR_PROBE)
export HNR_LIB_PATH=$PWD/hybridneuralrendering_amd/csrc/build/libhnr_probe.so
for d in ${S3_ABL:-0 1 2 4 8 16 3 11 15 31}; do
  HNR_S3_DBG=$d timeout 120 python tools/probe_s3.py --rows 4000000 2>&1 | grep "split-bf16:\|dbg32\|s3w" | tail -5 | sed "s/^/dbg=$d /"
done
