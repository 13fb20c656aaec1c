#!/bin/bash
# tools/build_variant2.sh NAME FILE.hip [extra flags]: the product's objects with csrc/FILE.hip recompiled with extra flags -> libhnr_hip_NAME.so
set -e
NAME=$1; F=$2; shift 2
cd "$(dirname "$0")/../hybridneuralrendering_amd/csrc"
make -s >/dev/null
mkdir -p build/variants/$NAME
B=$(basename $F .hip)
/opt/rocm/bin/hipcc "$@" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -I. -Wno-unused-variable -Wno-unused-but-set-variable -c $F -o build/variants/$NAME/$B.o
OBJS=$(ls build/*.o | grep -v "build/$B" | grep -v "amdgcn")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhnr_hip_$NAME.so $OBJS build/variants/$NAME/$B.o
echo built ../libhnr_hip_$NAME.so
