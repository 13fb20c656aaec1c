#!/bin/bash
# A/B builds of the chain kernel: tools/build_variant.sh NAME SRC.hip [extra hipcc flags] -> hybridneuralrendering_amd/libhnr_hip_NAME.so
# (the product's other objects + SRC compiled in place of csrc/chain_ws.hip).  Used with tools/ab_chain.py.
set -e
NAME=$1; SRC=$(realpath $2); shift 2
cd "$(dirname "$0")/../hybridneuralrendering_amd/csrc"
make -s >/dev/null
mkdir -p build/variants/$NAME
/opt/rocm/bin/hipcc "$@" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -I. -Wno-unused-variable -Wno-unused-but-set-variable \
    -c $SRC -o build/variants/$NAME/chain_ws.o
OBJS=$(ls build/*.o | grep -v "chain_ws\|-hip-amdgcn")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhnr_hip_$NAME.so $OBJS build/variants/$NAME/chain_ws.o
echo built ../libhnr_hip_$NAME.so
