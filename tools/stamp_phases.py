#!/usr/bin/env python3
"""Diagnostic: where does one wn_layer_kernel workgroup spend its cycles?  (GPU box only.)

Builds a -DWG_STAMPS copy of the library (the shipped one contains no stamp), runs one infer at the bench
shape and prints per-phase mean cycles over the workgroups of the LAST wn_layer launch with a residual
(has_res) -- shares only, the stamped build's run time is not a performance number.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "waveglow_amd", "csrc")
lib = os.path.join(ROOT, "gpurun_out", "libwaveglow_amd_stamps.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
extra = [f for f in sys.argv[2:] if f.startswith("-D") and f != "-DNONE"]
if os.environ.get("WG_STAMP_LIB"):          # prebuilt with tools/build_variant.sh <name> "-DWG_STAMPS ..."
  lib = os.path.abspath(os.environ["WG_STAMP_LIB"])
else:
  subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
                  "-DWG_STAMPS"] + extra + ["-o", lib, "kernels.hip", "stft.hip", "train.hip", "train_prep.hip", "api.cpp", "stft_api.cpp", "train_api.cpp"], cwd=csrc, check=True)
os.environ["WAVEGLOW_AMD_LIB"] = lib

import torch  # noqa: E402
from waveglow_amd import synthetic  # noqa: E402
from waveglow_amd.hparams import HParams  # noqa: E402
from waveglow_amd.model import WaveGlow  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B, T = (16, 864)
for f in sys.argv[2:]:
  if f.startswith("B="):
    B = int(f[2:])
  if f.startswith("T="):
    T = int(f[2:])
hp = HParams(n_channels=C, n_flows=2, n_early_every=4)   # 2 flows are enough for phase shares
# n_flows=2 leaves 8 channels; fine for timing
sd = synthetic.make_state_dict(hp, 0)
m = WaveGlow.remove_weightnorm(WaveGlow(hp))
m.load_state_dict(sd)
m = m.cuda().eval()
mel = synthetic.make_mel(B, T).cuda().half()
with torch.no_grad():
  m.infer(mel, 0.6)
  eng = m._engine
  BN = 64 if C >= 512 else 128
  if 32 * ((B * (T + 8) + 127) // 128) < torch.cuda.get_device_properties(0).multi_processor_count:
    BN = 64                                                  # small workloads: the library picks 64-column tiles (api.cpp: run_wn)
  n_tiles = 32 * ((B * (T + 8) + 127) // 128 * 128) // BN   # 32 phases x (rows per phase block / BN), see RowGeom
  buf = torch.zeros(n_tiles * 8, dtype=torch.int64, device="cuda")
  eng.lib.wg_debug_set_stamp_buffer(eng.handle, buf.data_ptr())
  m.infer(mel, 0.6)
  torch.cuda.synchronize()
st = buf.view(n_tiles, 8).cpu().double()
if os.environ.get("WG_STAMP_LAYOUT", "pipe") == "pipe":
  # pipelined epilogue: stamps 0 top, 1 K loop start, 2 K loop end, 3 / 5 / 6 / 7 end of phases 0..3, 4 end of the last phase
  order = [0, 1, 2, 3, 5, 6, 7, 4]
  names = ["prologue (bias, first DMA + A, barrier)", "K loop (GEMM1)", "phase 0 (gate 0, GEMM-2 weight loads)",
           "phase 1 (gate 1 || GEMM 2 of 0)", "phase 2", "phase 3", "phase 4 (GEMM 2 of 3, end x skip)"]
  if BN == 64:                                               # two column chunks: phases 0, 1, 2
    order = [0, 1, 2, 3, 5, 4]
    names = names[:4] + ["phase 2 (GEMM 2 of 1, end x skip)"]
else:
  order = [0, 1, 2, 3, 4, 5, 6]
  names = ["prologue (bias, first DMA + A, barrier)", "K loop (GEMM1)", "post-loop loads + gate + acts->LDS + barrier",
           "GEMM2 (res) incl. acc2 init", "folded end*skip + out RMW", "x_out stores"]
so = st[:, order]
d = so[:, 1:] - so[:, :-1]
tot = (so[:, -1] - so[:, 0]).mean().item()
print(f"C={C} tiles={n_tiles}: mean cycles per workgroup (stamps of the last launch that has a residual: flow 0, layer n_layers-2, dilation 64)")
for i, n in enumerate(names):
  print(f"  {n:48s} {d[:, i].mean().item():10.0f}  ({100 * d[:, i].mean().item() / tot:5.1f} %)")
print(f"  {'total':48s} {tot:10.0f}")
