"""Training-step timing on BASELINE config 4 shapes (per GPU: batch 32 x 16000 samples, 63 mel frames, fp32 I/O):
forward (saved activations) + WaveGlowLoss + backward (+ Adam), synthetic data / random-init weights.

  python tools/bench_train.py [--batch 32] [--steps 5] [--warmup 2] [--channels 256] [--adam]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from waveglow_amd import synthetic  # noqa: E402
from waveglow_amd.hparams import HParams  # noqa: E402
from waveglow_amd.model import WaveGlow, WaveGlowLoss  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--batch", type=int, default=32)
  ap.add_argument("--segment", type=int, default=16000)
  ap.add_argument("--steps", type=int, default=5)
  ap.add_argument("--warmup", type=int, default=2)
  ap.add_argument("--channels", type=int, default=256)
  ap.add_argument("--adam", action="store_true")
  a = ap.parse_args()
  hp = HParams(n_channels=a.channels)
  sd = synthetic.to_weightnorm_form(synthetic.make_state_dict(hp, seed=0))
  model = WaveGlow(hp)
  model.load_state_dict(sd)
  model = model.to("cuda:0").train()
  F_ = 1 + a.segment // 256
  mel = synthetic.make_mel(a.batch, F_, seed=7).cuda()
  g = torch.Generator().manual_seed(3)
  wav = (torch.rand(a.batch, a.segment, generator=g) * 0.6 - 0.3).cuda()
  crit = WaveGlowLoss(1.0)
  opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True) if a.adam else None
  ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
  t_f = t_b = t_o = 0.0
  wall = host = 0.0
  for it in range(a.warmup + a.steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.zero_grad(set_to_none=True)
    ev[0].record()
    y = model((mel, wav))
    loss = crit(y, None)
    ev[1].record()
    loss.backward()
    ev[2].record()
    if opt is not None:
      opt.step()
    ev[3].record()
    t_host = time.perf_counter() - t0            # everything queued; the device may still be busy
    torch.cuda.synchronize()
    if it >= a.warmup:
      host += t_host
      wall += time.perf_counter() - t0
      t_f += ev[0].elapsed_time(ev[1]); t_b += ev[1].elapsed_time(ev[2]); t_o += ev[2].elapsed_time(ev[3])
  n = a.steps
  samples = a.batch * a.segment
  macs = 81235408 / 8 if a.channels == 256 else None    # forward MACs per sample
  out = {"workload": f"train step WaveGlow-{a.channels} batch {a.batch} x {a.segment} samples", "loss": float(loss.detach()),
         "ms_forward": t_f / n, "ms_backward": t_b / n, "ms_optimizer": t_o / n, "ms_step_wall": 1e3 * wall / n, "ms_host_enqueue": 1e3 * host / n,
         "samples_per_s": samples / (wall / n), "mem_GiB": torch.cuda.max_memory_allocated() / 2**30}
  if macs:
    out["algorithmic_TFLOPs_fwd_plus_bwd"] = 3 * 2 * macs * samples / 1e12
    out["TFLOP_per_s"] = out["algorithmic_TFLOPs_fwd_plus_bwd"] / (wall / n)
  print(json.dumps(out))


if __name__ == "__main__":
  main()
