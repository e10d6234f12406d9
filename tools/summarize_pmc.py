#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs (one dir per pass) into per-kernel averages per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
name_chars = int(sys.argv[2]) if len(sys.argv) > 2 else 48     # the training step needs ~76 to tell the wn_layer_kernel modes apart
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
  with open(path) as f:
    for row in csv.DictReader(f):
      name = row.get("Kernel_Name", "")[:name_chars]
      c = row.get("Counter_Name")
      v = float(row.get("Counter_Value", 0))
      a = agg[name][c]
      a[0] += v
      a[1] += 1
for name in sorted(agg):
  if "wg::" not in name:
    continue
  print(name)
  for c in sorted(agg[name]):
    s, n = agg[name][c]
    print(f"   {c:36s} avg/dispatch {s / n:16.1f}   dispatches {n}")
