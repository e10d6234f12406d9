#!/bin/bash
# Build a variant of the library with extra -D flags into gpurun_out/ and run a pytest selection against it.
# usage: tools/test_variant.sh "<-D flags>" <pytest -k expression>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
FLAGS=$1; shift
LIB=$ROOT/gpurun_out/lib_variant.so
mkdir -p $ROOT/gpurun_out
(cd $ROOT/waveglow_amd/csrc && hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -Wno-unused-value $FLAGS -o $LIB kernels.hip stft.hip train.hip train_prep.hip api.cpp stft_api.cpp train_api.cpp) || exit 1
WAVEGLOW_AMD_LIB=$LIB timeout -k 10 300 python -m pytest $ROOT/tests -m gpu -x -q -k "$*" 2>&1 | tail -4
