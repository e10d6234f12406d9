"""Golden GRADIENTS from the reference's own training step (build container only; imports the reference).

  python tests/golden/make_golden_grads.py

For the small cases: WaveGlow.forward (model.py:178-221, weight-normed parameters) -> WaveGlowLoss (train.py:31-45)
-> loss.backward() on CPU fp32, exactly the reference's train.py:190-196 sequence.  Stored per parameter: the
gradient's L2 norm, its sum, and the first 8 values -- plus full gradients for the small tensors (convinv, start,
end, biases), so the backward kernels of the training direction (not built in round 1) can be pinned later.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from _ref_import import import_reference  # noqa: E402
from waveglow_amd.hparams import HParams  # noqa: E402
from waveglow_amd import synthetic  # noqa: E402

ref_model, ref_hparams, ref_train = import_reference()

# name: (hparam overrides, B, mel frames T, weight seed, audio samples S or None for 256*T - 96)
CASES = {"c64": (dict(n_channels=64, n_layers=4, n_flows=6, n_early_every=2), 2, 12, 5, None),
         # BASELINE configs[3] shapes (256 channels, 16 000-sample segments, 63 mel frames) at the batch the reference's
         # CPU probe used (B = 2): summaries only
         "cfg4_b2": (dict(), 2, 63, 0, 16000)}
if len(sys.argv) > 1:
  CASES = {k: v for k, v in CASES.items() if k in sys.argv[1:]}

for name, (over, B, T, wseed, S_) in CASES.items():
  hp = HParams(**over)
  sd = synthetic.make_state_dict(hp, seed=wseed)
  model = ref_model.WaveGlow(ref_hparams.HParams(**over))
  model.load_state_dict(synthetic.to_weightnorm_form(sd))
  model.train()
  mel = synthetic.make_mel(B, T, seed=1234 + B + T)
  S = S_ if S_ is not None else 256 * T - 96
  g = torch.Generator().manual_seed(99 + T)
  wav = torch.rand(B, S, generator=g) * 0.6 - 0.3
  model.zero_grad()
  y = model((mel, wav))
  loss = ref_train.WaveGlowLoss(sigma=1.0)(y, None)
  loss.backward()
  out = {"loss": np.array(float(loss), dtype=np.float32)}
  for pname, p in model.named_parameters():
    gr = p.grad.detach()
    out["norm/" + pname] = np.array(float(gr.norm()), dtype=np.float32)
    out["sum/" + pname] = np.array(float(gr.sum()), dtype=np.float32)
    out["head/" + pname] = gr.flatten()[:8].numpy().copy()
    if gr.numel() <= 4096 and name == "c64":
      out["full/" + pname] = gr.numpy().copy()
  np.savez_compressed(os.path.join(HERE, f"{name}_grads.npz"), **out)
  print(name, "loss", float(loss), "params", sum(1 for _ in model.named_parameters()))
