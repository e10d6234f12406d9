#!/bin/bash
# Interleaved bench of the in-tree library (A) and prebuilt variant libraries on ONE box:
#   tools/ab_libs.sh build_variants/lib_a.so build_variants/lib_b.so ...      (build them with tools/build_variant.sh)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
ROUNDS=${ROUNDS:-3}
for round in $(seq 1 $ROUNDS); do
  for v in "" "$@"; do
    if [ -z "$v" ]; then unset WAVEGLOW_AMD_LIB; name="A (in-tree)"; else export WAVEGLOW_AMD_LIB=$ROOT/$v; name=$(basename $v .so); fi
    timeout -k 10 300 python $ROOT/bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 3 $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
  if l.startswith('{'):
    d = json.loads(l); r = d['roofline']
    print('$name round $round: %.3f ms/step  kernel avg %.4f ms  frac %.4f' % (d['ms_per_step'], r['avg_launch_ms'], r['frac']))
"
  done
done
