"""Diagnostic (needs a -DWGR_STAMPS build, WAVEGLOW_AMD_LIB): per-workgroup timeline of the LAST wgrad launch of a
training step at configs[3] shapes.  Stamps: 0 kernel entry, 1 before the first DMA, 2 top of step 1, 3 top of the middle
step, 4 loop end, 5 stores drained."""
import os
import sys

import numpy as np
import torch

os.environ["WG_TRAIN_SERIAL"] = "1"
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from waveglow_amd import synthetic  # noqa: E402
from waveglow_amd.hparams import HParams  # noqa: E402
from waveglow_amd.model import WaveGlow, WaveGlowLoss  # noqa: E402

hp = HParams()
model = WaveGlow(hp)
model.load_state_dict(synthetic.to_weightnorm_form(synthetic.make_state_dict(hp, seed=0)))
model = model.cuda().train()
B, S = 32, 16000
mel = synthetic.make_mel(B, 1 + S // 256, seed=77).cuda()
wav = (torch.rand(B, S, generator=torch.Generator().manual_seed(5)) * 0.6 - 0.3).cuda()
crit = WaveGlowLoss(1.0)
for _ in range(2):
  model.zero_grad(set_to_none=True)
  crit(model((mel, wav)), None).backward()
eng = model._engine
buf = torch.zeros(1024 * 32, dtype=torch.int64, device="cuda")
eng.lib.wg_debug_set_stamp_buffer(eng.handle, buf.data_ptr())
model.zero_grad(set_to_none=True)
crit(model((mel, wav)), None).backward()
torch.cuda.synchronize()
st = buf.view(1024, 32).cpu().numpy().astype(np.int64)
st = st[st[:, 0] != 0]
print("workgroups stamped", len(st))
t0 = st[:, 0].min()
tile = (st[:, 6] >> 32).astype(int)
slab = (st[:, 6] & 0xffffffff).astype(int)
nst = (st[:, 7] >> 32).astype(int)
rel = st[:, :6] - t0
print("kernel span (cycles, s_memtime = 100 MHz ticks x ?):", int(rel[:, 5].max()))
for j, name in enumerate(["entry", "before first DMA", "top of step 1", "top of middle step", "loop end", "stores drained"]):
  c = rel[:, j]
  print(f"{name:20s} min {c.min():9d} median {int(np.median(c)):9d} max {c.max():9d}")
dur = rel[:, 4] - rel[:, 1]
per_step = dur / nst
print("loop cycles per step: min %.0f median %.0f max %.0f" % (per_step.min(), np.median(per_step), per_step.max()))
for t in sorted(set(tile)):
  m = tile == t
  print(f"tile {t:2d}: n {m.sum():3d} per-step median {np.median(per_step[m]):7.0f} max {per_step[m].max():7.0f}  epilogue median {np.median((rel[:, 5] - rel[:, 4])[m]):7.0f}"
        f"  first-half/second-half per step {np.median(((rel[:, 3] - rel[:, 2]) / np.maximum(nst // 2 - 1, 1))[m]):7.0f} / {np.median(((rel[:, 4] - rel[:, 3]) / np.maximum(nst - nst // 2, 1))[m]):7.0f}")

names = ["top", "vmcnt done", "barrier passed", "DMA issued", "late MFMA done", "reads issued", "early MFMA done"]
for label, o in (("wave 0 (early half)", 8), ("wave 4 (late half)", 16)):
  d = st[:, o:o + 7]
  ok = (d[:, 0] != 0) & (tile >= 2) & (tile <= 9)
  print(label, "step 10, tiles 2-9, cycles since the step's top (median / p90):")
  for j in range(1, 7):
    c = (d[ok, j] - d[ok, 0])
    print(f"   {names[j]:18s} {int(np.median(c)):6d} / {int(np.percentile(c, 90)):6d}")
