"""Build libwaveglow_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(CSRC, "libwaveglow_amd.so")
SOURCES = ["kernels.hip", "stft.hip", "train.hip", "train_prep.hip", "api.cpp", "stft_api.cpp", "train_api.cpp"]
HEADERS = ["wg_common.h", "wg_train.h", os.path.join("..", "..", "include", "waveglow_amd.h")]


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
  """hipcc -> in-tree .so.  The build FAILS if any kernel spills registers or uses scratch: the WN-layer
  kernel issues loads from inline asm with hand-counted waits, and a compiler spill of such a register
  (a scratch store before the data has landed) would silently corrupt results."""
  if not force and not needs_build():
    return LIB
  cmd = [_hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
         "-Rpass-analysis=kernel-resource-usage", "-o", LIB + ".tmp"] + SOURCES
  res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
  if res.returncode != 0:
    raise RuntimeError("hipcc failed:\n" + res.stderr)
  report, name = [], None
  for line in res.stderr.splitlines():
    if "Function Name:" in line:
      name = line.split("Function Name:")[1].split("[")[0].strip()
    # SGPR spills go to VGPR lanes (v_writelane), never to memory: tolerated.  VGPR spills / scratch are not.
    for key in ("VGPRs Spill:", "ScratchSize [bytes/lane]:"):
      if key in line and name:
        val = int(line.split(key)[1].split("[")[0].strip())
        if val != 0:
          report.append(f"{name}: {key} {val}")
  if verbose:
    print(res.stderr)
  if report:
    os.remove(LIB + ".tmp")
    raise RuntimeError("register spills / scratch in device code (forbidden, see build_library doc):\n  " + "\n  ".join(report))
  os.replace(LIB + ".tmp", LIB)
  return LIB
