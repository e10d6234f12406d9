#!/bin/bash
# A/B the in-tree library (A) against a build of another source directory (B) on ONE box, interleaved.
# usage: tools/ab_bench_src.sh <dir with the csrc of variant B> [bench args]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
SRC=$1; shift
LIB=$ROOT/gpurun_out/lib_variantB.so
mkdir -p $ROOT/gpurun_out
(cd $SRC && hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -Wno-unused-value -o $LIB kernels.hip stft.hip train.hip train_prep.hip api.cpp stft_api.cpp train_api.cpp) || exit 1
for round in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then unset WAVEGLOW_AMD_LIB; else export WAVEGLOW_AMD_LIB=$LIB; fi
    timeout -k 10 300 python $ROOT/bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
  if l.startswith('{'):
    d = json.loads(l); r = d['roofline']
    print('$v round $round: %.3f ms/step  wn_layer avg %.4f ms  frac %.4f' % (d['ms_per_step'], r['avg_launch_ms'], r['frac']))
"
  done
done
