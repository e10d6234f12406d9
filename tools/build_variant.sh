#!/bin/bash
# Build a variant of the library with extra -D flags (CPU container, cross-compile): tools/build_variant.sh <name> "<flags>"
# -> build_variants/lib_<name>.so (travels to the GPU box with the snapshot; *.so is git-ignored).  Fails on VGPR spills.
ROOT=$(cd $(dirname $0)/.. && pwd)
mkdir -p $ROOT/build_variants
cd $ROOT/waveglow_amd/csrc
hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -Wno-unused-value -Rpass-analysis=kernel-resource-usage $2 \
  -o $ROOT/build_variants/lib_$1.so kernels.hip stft.hip train.hip train_prep.hip api.cpp stft_api.cpp train_api.cpp 2> /tmp/build_$1.err || { grep -v remark /tmp/build_$1.err | head -30; exit 1; }
if grep "VGPRs Spill\|ScratchSize" /tmp/build_$1.err | grep -qv ": 0 "; then echo "variant $1: spills / scratch"; exit 2; fi
echo "built build_variants/lib_$1.so"
