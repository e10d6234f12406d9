"""Debug aid: one training step at a small shape with a NaN-poisoned gradient buffer; reports which views of the flat
buffer hold non-finite values and the per-parameter error against the CPU oracle.  usage: debug_grads.py [channels] [B] [T]"""
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["WG_TRAIN_POISON_GRADS"] = "1"
from _cases import oracle_cfg_from_hp  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402
import importlib  # noqa: E402
from waveglow_amd import synthetic  # noqa: E402
T_ = importlib.import_module("waveglow_amd.train")
from waveglow_amd.hparams import HParams  # noqa: E402
from waveglow_amd.model import WaveGlow, WaveGlowLoss  # noqa: E402

ch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
T = int(sys.argv[3]) if len(sys.argv) > 3 else 6
hp = HParams(n_channels=ch, n_layers=3, n_flows=4, n_early_every=2)
sd = synthetic.to_weightnorm_form(synthetic.make_state_dict(hp, seed=2))
mel = synthetic.make_mel(B, T, seed=1234 + B + T)
wav = torch.rand(B, 256 * T - 96, generator=torch.Generator().manual_seed(99 + T)) * 0.6 - 0.3

last = {}
_orig = T_.GradBuffers.__init__


def _init(self, *a, **k):
  _orig(self, *a, **k)
  last["bufs"] = self


T_.GradBuffers.__init__ = _init
model = WaveGlow(hp)
model.load_state_dict(sd)
model = model.cuda().train()
loss = WaveGlowLoss(1.0)(model((mel.cuda(), wav.cuda())), None)
loss.backward()
torch.cuda.synchronize()
b = last["bufs"]
print("grad_finite", bool(model.grad_finite), "loss", float(loss))
for name in ("dw1", "db1", "dw2", "db2", "dwes", "dstart", "dout_init", "dw1x1", "dwup", "dbup"):
  t = getattr(b, name)
  bad = ~torch.isfinite(t)
  print(f"{name:10s} shape {tuple(t.shape)} non-finite {int(bad.sum())}")
  if bad.any():
    idx = bad.nonzero()
    print("   first", idx[:4].tolist(), "last", idx[-2:].tolist())
    for d in range(idx.shape[1]):
      print(f"   dim {d}: distinct {sorted(set(idx[:, d].tolist()))[:24]}")
loss_ref, g_ref = O.grads_ref(sd, mel, wav, oracle_cfg_from_hp(hp), 1.0)
worst = []
for n, p in model.named_parameters():
  g = p.grad.detach().float().cpu()
  worst.append((float((g - g_ref[n]).norm()) / max(float(g_ref[n].norm()), 1e-12), n))
worst.sort(reverse=True)
for rel, n in worst[:12]:
  print(f"{rel:.3e} {n}")
