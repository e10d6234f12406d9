#!/bin/bash
# Interleaved training-step bench of the in-tree library (A) and prebuilt variant libraries on ONE box.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
ROUNDS=${ROUNDS:-2}
for round in $(seq 1 $ROUNDS); do
  for v in "" "$@"; do
    if [ -z "$v" ]; then unset WAVEGLOW_AMD_LIB; name="A (in-tree)"; else export WAVEGLOW_AMD_LIB=$ROOT/$v; name=$(basename $v .so); fi
    timeout -k 10 300 python $ROOT/tools/bench_train.py --adam --steps 8 --warmup 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
  if l.startswith('{'):
    d = json.loads(l)
    print('$name round $round:', ' '.join('%s %.2f' % (k, v) for k, v in d.items() if k.startswith('ms_')))
"
  done
done
