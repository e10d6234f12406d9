"""Single-utterance latency (BASELINE configs[0] shape: 256 ch, mel [1,80,500], fp32): direct launches vs hipGraph replay."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from waveglow_amd import synthetic  # noqa: E402
from waveglow_amd.hparams import HParams  # noqa: E402
from waveglow_amd.model import WaveGlow  # noqa: E402

hp = HParams()
model = WaveGlow.remove_weightnorm(WaveGlow(hp))
model.load_state_dict(synthetic.make_state_dict(hp, seed=0))
model = model.cuda().eval()
T = 500
mel = synthetic.make_mel(1, T).cuda()
z_init, z_early = synthetic.make_noise(hp, 1, 32 * T)
zi = z_init.cuda()
ze = [z_early[k].cuda() for k in sorted(z_early, reverse=True)]
out = {}
with torch.no_grad():
  for name, flag in (("direct", False), ("graph", True)):
    for _ in range(5):
      model.infer_with_noise(mel, zi, ze, 0.6, graph=flag)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
      model.infer_with_noise(mel, zi, ze, 0.6, graph=flag)
    torch.cuda.synchronize()
    out[name + "_ms"] = 1e3 * (time.perf_counter() - t0) / n
out["samples"] = 256 * T
out["samples_per_s_graph"] = 256 * T / (out["graph_ms"] * 1e-3)
print(json.dumps(out))
