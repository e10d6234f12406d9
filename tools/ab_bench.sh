#!/bin/bash
# A/B two builds of the library on ONE box, interleaved (rule 24): tools/ab_bench.sh "<-D flags of variant B>" [bench args]
# A = the in-tree library, B = the same sources with the extra flags.  Prints ms/step + avg wn_layer launch per run.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
FLAGS=$1; shift
LIB=$ROOT/gpurun_out/lib_variantB.so
mkdir -p $ROOT/gpurun_out
(cd $ROOT/waveglow_amd/csrc && hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -Wno-unused-value $FLAGS -o $LIB kernels.hip stft.hip train.hip train_prep.hip api.cpp stft_api.cpp train_api.cpp) || exit 1
for round in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then unset WAVEGLOW_AMD_LIB; else export WAVEGLOW_AMD_LIB=$LIB; fi
    timeout -k 10 300 python $ROOT/bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
  if l.startswith('{'):
    d = json.loads(l); r = d['roofline']
    print('$v round $round: %.3f ms/step  wn_layer avg %.4f ms  frac %.4f  value %.1f' % (d['ms_per_step'], r['avg_launch_ms'], r['frac'], d['value']))
"
  done
done
