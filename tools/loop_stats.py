"""Instruction mix of every MFMA-carrying loop of a kernel in hipcc's -S output: tools/loop_stats.py file.s [kernel-substring]"""
import re
import sys

text = open(sys.argv[1]).read().split("\n")
key = sys.argv[2] if len(sys.argv) > 2 else ""
labels = {}
for i, l in enumerate(text):
  m = re.match(r"^(\.LBB\d+_\d+):", l)
  if m:
    labels[m.group(1)] = i
pats = {"mfma": "v_mfma", "ds_read": "ds_read", "dma": "global_load_lds|buffer_load.*lds", "vmem": r"^\s*(global|buffer)_(load|store)",
        "branches": r"s_c?branch", "salu": r"^\s*s_", "valu": r"^\s*v_", "lanes": "v_readlane|v_writelane", "waitcnt": "s_waitcnt",
        "barrier": "s_barrier"}
for i, l in enumerate(text):
  m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)\s*$", l)
  if not m or m.group(1) not in labels or labels[m.group(1)] >= i:
    continue
  body = text[labels[m.group(1)]:i + 1]
  cnt = {k: sum(bool(re.search(p, b)) for b in body) for k, p in pats.items()}
  if cnt["mfma"] >= 8:
    print(f"loop lines {labels[m.group(1)]}..{i}: " + " ".join(f"{k} {v}" for k, v in cnt.items()))
