"""Diagnostic: which torch ops (by input shape) run around the library calls of one training step -- the "plumbing" that
packs weights and hands every parameter its gradient.  GPU box only: python tools/profile_train_ops.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from waveglow_amd import synthetic  # noqa: E402
from waveglow_amd.hparams import HParams  # noqa: E402
from waveglow_amd.model import WaveGlow, WaveGlowLoss  # noqa: E402

hp = HParams()
model = WaveGlow(hp)
model.load_state_dict(synthetic.to_weightnorm_form(synthetic.make_state_dict(hp, seed=0)))
model = model.to("cuda:0").train()
mel = synthetic.make_mel(32, 63, seed=7).cuda()
wav = (torch.rand(32, 16000, generator=torch.Generator().manual_seed(3)) * 0.6 - 0.3).cuda()
crit = WaveGlowLoss(1.0)
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)


def step():
  model.zero_grad(set_to_none=True)
  loss = crit(model((mel, wav)), None)
  loss.backward()
  opt.step()


for _ in range(3):
  step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
  step()
  torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
print(ka.table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=40, max_shapes_column_width=70))
print("---- copy-type ops by shape (count, device time in us, name, shapes)")
rows = [e for e in ka if e.key in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zeros", "aten::zero_", "aten::cat",
                                   "aten::stack", "aten::index_select", "aten::add", "aten::mul", "aten::constant_pad_nd", "aten::sum")]
rows.sort(key=lambda e: -e.count)
for e in rows[:60]:
  print("%5d %9.1f  %-22s %s" % (e.count, getattr(e, "device_time_total", getattr(e, "cuda_time_total", 0.0)), e.key, str(e.input_shapes)[:150]))

print("---- every op / kernel with >= 40 calls")
for e in sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.count):
  if e.count >= 40:
    print("%5d %9.1f  %-60s %s" % (e.count, getattr(e, "device_time_total", 0.0), e.key[:60], str(e.input_shapes)[:120]))
