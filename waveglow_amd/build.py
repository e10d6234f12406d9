"""Build libwaveglow_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(CSRC, "libwaveglow_amd.so")
SOURCES = ["kernels.hip", "api.cpp"]
HEADERS = ["wg_common.h", os.path.join("..", "..", "include", "waveglow_amd.h")]


def _hipcc() -> str:
  for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
    if cand and os.path.isfile(cand):
      return cand
  raise RuntimeError("hipcc not found")


def needs_build() -> bool:
  if not os.path.isfile(LIB):
    return True
  t = os.path.getmtime(LIB)
  return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_library(force: bool = False, verbose: bool = False) -> str:
  if not force and not needs_build():
    return LIB
  cmd = [_hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC",
         "-Wno-unused-value", "-o", LIB] + SOURCES
  if verbose:
    cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
  subprocess.run(cmd, cwd=CSRC, check=True)
  return LIB
