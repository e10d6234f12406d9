#!/bin/bash
# Throughput over the batch size at 80x864 mels (and, with FORCE="64 128", both tile widths): tools/sweep_batch.sh "1 2 4 8 16 32"
for b in ${1:-1 2 4 8 16 32}; do
  for f in ${FORCE:-auto}; do
    if [ "$f" = auto ]; then unset WG_FORCE_BN; else export WG_FORCE_BN=$f; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 5 --warmup 2 --batch $b 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
  if l.startswith('{'):
    d = json.loads(l); r = d['roofline']
    print('B=$b tile=$f: %.3f ms/step  %.1f M samples/s  kernel avg %.4f ms  frac %.4f' % (d['ms_per_step'], d['value']/1e6, r['avg_launch_ms'], r['frac']))
"
  done
done
