#!/bin/bash
# Probe build of the library (-DHNR_LINEAR_PROBE: ablation instantiations of the dense-layer kernels) -> csrc/build/libhnr_probe.so
set -e
cd "$(dirname "$0")/../hybridneuralrendering_amd/csrc"
make -s
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $F -DHNR_LINEAR_PROBE -c linear_s3.hip -o build/probe_linear_s3.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libhnr_probe.so $(ls build/*.o | grep -v "build/linear_s3.o\|probe_linear_s3") build/probe_linear_s3.o
