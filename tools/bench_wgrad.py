"""wgrad_kernel on its own at BASELINE configs[3] shapes: a few training steps with every launch on one stream
(WG_TRAIN_SERIAL=1) and hipEvents around the weight-gradient launches only; no optimiser (ablation builds compute
garbage).  Prints the average launch time.  WAVEGLOW_AMD_LIB selects a variant library."""
import ctypes as C
import json
import os
import sys

import torch

os.environ["WG_TRAIN_SERIAL"] = "1"
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from waveglow_amd import synthetic  # noqa: E402
from waveglow_amd.hparams import HParams  # noqa: E402
from waveglow_amd.model import WaveGlow, WaveGlowLoss  # noqa: E402

ch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
hp = HParams(n_channels=ch)
model = WaveGlow(hp)
model.load_state_dict(synthetic.to_weightnorm_form(synthetic.make_state_dict(hp, seed=0)))
model = model.cuda().train()
S = 16000
mel = synthetic.make_mel(B, 1 + S // 256, seed=77).cuda()
wav = (torch.rand(B, S, generator=torch.Generator().manual_seed(5)) * 0.6 - 0.3).cuda()
crit = WaveGlowLoss(1.0)


def step():
  model.zero_grad(set_to_none=True)
  crit(model((mel, wav)), None).backward()


step()
eng = model._engine
torch.cuda.synchronize()
eng.lib.wg_profile_enable(eng.handle, (1 << 5) | (1 << 6))
n = 3
for _ in range(n):
  step()
torch.cuda.synchronize()
ms = (C.c_double * 8)()
cnt = (C.c_int64 * 8)()
eng.lib.wg_profile_read(eng.handle, ms, cnt, 8)
print(json.dumps({"lib": os.environ.get("WAVEGLOW_AMD_LIB", "in-tree"), "wgrad_avg_launch_us": round(1e3 * ms[6] / max(1, cnt[6]), 2),
                  "wgrad_ms_per_step": round(ms[6] / n, 3), "dgrad_ms_per_step": round(ms[5] / n, 3), "launches": int(cnt[6])}))
