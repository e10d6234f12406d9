"""Golden SUMMARY of BASELINE configs[0] from the reference's own model code (build container only; SURVEY 8c, plan item 3).

  python tests/golden/make_golden_cfg1.py

256 channels, mel [1, 80, 500] -> 128 000 samples: too large to store whole, so the fixture keeps the inputs'
generator seeds, summary statistics of the audio, its first / last 256 samples and 1024 samples at a fixed stride.
The mel and the weights are regenerated from their seeds at test time (waveglow_amd.synthetic; the weights' crc32 guards
generator drift); the noise is what ``WaveGlow.infer`` itself drew after ``torch.manual_seed(noise_seed)`` -- the test
replays those draws (torch.FloatTensor(...).normal_() in the reference's order) and injects them.
"""
import os
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from _ref_import import import_reference  # noqa: E402
from waveglow_amd.hparams import HParams  # noqa: E402
from waveglow_amd import synthetic  # noqa: E402

ref_model, ref_hparams, ref_train = import_reference()
torch.set_num_threads(8)
hp = HParams()
sd = synthetic.make_state_dict(hp, seed=0)
model = ref_model.WaveGlow.remove_weightnorm(ref_model.WaveGlow(ref_hparams.HParams()))
model.load_state_dict(sd)
model = model.eval()
B, T, sigma, mel_seed, noise_seed = 1, 500, 0.6, 1234, 4321
mel = synthetic.make_mel(B, T, seed=mel_seed)
with torch.no_grad():
  torch.manual_seed(noise_seed)                    # the reference draws its noise from the global CPU RNG (model.py:234-271)
  audio = model.infer(mel, sigma=sigma)
a = audio[0].double()
crc = 0
for key in sorted(sd):
  crc = zlib.crc32(sd[key].numpy().tobytes(), crc)
idx = np.arange(0, a.numel(), a.numel() // 1024)[:1024]
np.savez_compressed(os.path.join(HERE, "cfg1_summary.npz"), T=np.array(T), sigma=np.array(sigma, dtype=np.float32),
                    mel_seed=np.array(mel_seed), noise_seed=np.array(noise_seed), weight_seed=np.array(0),
                    weights_crc32=np.array(crc, dtype=np.uint32), n_samples=np.array(a.numel()),
                    mean=np.array(float(a.mean())), rms=np.array(float(a.pow(2).mean().sqrt())),
                    max_abs=np.array(float(a.abs().max())), first=audio[0, :256].numpy(), last=audio[0, -256:].numpy(),
                    strided_index=idx, strided=audio[0].numpy()[idx])
print("cfg1: samples", a.numel(), "rms", float(a.pow(2).mean().sqrt()), "max", float(a.abs().max()), "crc", hex(crc))
