#!/bin/bash
# Interleaved bench of the in-tree library (A) and several -D variants on ONE box: tools/ab_multi.sh "<flags1>" "<flags2>" ...
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
i=0
for F in "$@"; do
  i=$((i+1))
  (cd $ROOT/waveglow_amd/csrc && hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -Wno-unused-value $F -o $ROOT/gpurun_out/lib_v$i.so kernels.hip stft.hip train.hip train_prep.hip api.cpp stft_api.cpp train_api.cpp) || exit 1
done
for round in 1 2 3; do
  for v in $(seq 0 $i); do
    if [ $v = 0 ]; then unset WAVEGLOW_AMD_LIB; name="A (in-tree)"; else export WAVEGLOW_AMD_LIB=$ROOT/gpurun_out/lib_v$v.so; name="V$v"; fi
    timeout -k 10 300 python $ROOT/bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
  if l.startswith('{'):
    d = json.loads(l); r = d['roofline']
    print('$name round $round: %.3f ms/step  wn_layer avg %.4f ms  frac %.4f' % (d['ms_per_step'], r['avg_launch_ms'], r['frac']))
"
  done
done
