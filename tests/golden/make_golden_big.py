"""Golden SUMMARIES of the large BASELINE configurations from the reference's own model code (build container only).

  python tests/golden/make_golden_big.py [cfg5] [cfg3] [c256wn]

  cfg5    BASELINE configs[4] per-utterance shape: 256 channels, mel [1, 80, 4000] -> 1 024 000 samples, sigma 0.6
  cfg3    BASELINE configs[2] per-utterance shape: 512 channels, mel [1, 80, 864]  ->   221 184 samples, sigma 0.6
  c256wn  256-channel weight-normed (686-key) checkpoint with g != ||v||, folded by the reference's own
          remove_weightnorm (model.py:276-297), mel [1, 80, 8]: full audio

The outputs are too large to store whole, so (as make_golden_cfg1.py does) a fixture keeps the generator seeds, summary
statistics, the first / last 256 samples and 2048 samples at a fixed stride.  Both configurations are fp16 I/O on the
GPU, so the inputs are made fp16-representable BEFORE the reference sees them: the mel is rounded through fp16, and the
three noise draws of ``WaveGlow.infer`` (model.py:234-244, :260-271) are rounded through fp16 right after the reference
draws them (``torch.Tensor.normal_`` is wrapped for the duration of the call -- a patch of torch in this process, not
of the reference).  "Identical mel + noise" for the fp16 GPU path then means exactly these values; the test replays
the draws (``torch.FloatTensor(...).normal_().half()``) and injects them.
"""
import os
import sys
import time
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


class fp16_noise:
  """Every ``Tensor.normal_()`` inside the block is rounded through fp16 in place (values stay fp32)."""

  def __enter__(self):
    self.orig = torch.Tensor.normal_
    orig = self.orig

    def normal_(t, *a, **k):
      orig(t, *a, **k)
      t.copy_(t.half().float())
      return t
    torch.Tensor.normal_ = normal_

  def __exit__(self, *exc):
    torch.Tensor.normal_ = self.orig


def crc_of(sd):
  crc = 0
  for key in sorted(sd):
    crc = zlib.crc32(sd[key].numpy().tobytes(), crc)
  return crc


def summary(name, hp_over, T, wseed, mel_seed, noise_seed, sigma=0.6, n_strided=2048):
  hp = HParams(**hp_over)
  sd = synthetic.make_state_dict(hp, seed=wseed)
  model = ref_model.WaveGlow.remove_weightnorm(ref_model.WaveGlow(ref_hparams.HParams(**hp_over)))
  model.load_state_dict(sd)
  model = model.eval()
  mel = synthetic.make_mel(1, T, seed=mel_seed).half().float()
  t0 = time.time()
  with torch.no_grad(), fp16_noise():
    torch.manual_seed(noise_seed)
    audio = model.infer(mel, sigma=sigma)
  a = audio[0].double()
  idx = np.arange(0, a.numel(), a.numel() // n_strided)[:n_strided]
  np.savez_compressed(os.path.join(HERE, f"{name}_summary.npz"), T=np.array(T), sigma=np.array(sigma, dtype=np.float32),
                      hp_json=np.array(repr(sorted(hp_over.items()))),
                      mel_seed=np.array(mel_seed), noise_seed=np.array(noise_seed), weight_seed=np.array(wseed),
                      weights_crc32=np.array(crc_of(sd), dtype=np.uint32), n_samples=np.array(a.numel()),
                      mean=np.array(float(a.mean())), rms=np.array(float(a.pow(2).mean().sqrt())),
                      max_abs=np.array(float(a.abs().max())), first=audio[0, :256].numpy(), last=audio[0, -256:].numpy(),
                      strided_index=idx, strided=audio[0].numpy()[idx])
  print(f"{name}: samples {a.numel()} rms {float(a.pow(2).mean().sqrt()):.5f} max {float(a.abs().max()):.4f} "
        f"crc {crc_of(sd):#x}  ({time.time() - t0:.1f} s)")


def c256_weightnorm():
  """686-key checkpoint whose g differs from ||v|| (so the fold g*v/||v|| is not the identity), folded and run by the
  reference itself."""
  hp = HParams()
  sd = synthetic.make_state_dict(hp, seed=9)
  wn = synthetic.to_weightnorm_form(sd)
  gen = torch.Generator().manual_seed(31)
  for key in sorted(wn):
    if key.endswith("original0"):
      wn[key] = wn[key] * (0.5 + torch.rand(wn[key].shape, generator=gen))
  model = ref_model.WaveGlow(ref_hparams.HParams())
  model.load_state_dict(wn)
  model = ref_model.WaveGlow.remove_weightnorm(model).eval()
  B, T, sigma, noise_seed = 1, 8, 0.6, 77
  mel = synthetic.make_mel(B, T, seed=55)
  with torch.no_grad():
    torch.manual_seed(noise_seed)
    audio = model.infer(mel, sigma=sigma)
  np.savez_compressed(os.path.join(HERE, "c256_wn.npz"), T=np.array(T), sigma=np.array(sigma, dtype=np.float32),
                      weight_seed=np.array(9), g_seed=np.array(31), mel_seed=np.array(55), noise_seed=np.array(noise_seed),
                      weights_crc32=np.array(crc_of(sd), dtype=np.uint32), audio=audio.numpy())
  print("c256wn: audio", tuple(audio.shape), "rms", float(audio.pow(2).mean().sqrt()))


if __name__ == "__main__":
  which = sys.argv[1:] or ["c256wn", "cfg3", "cfg5"]
  if "c256wn" in which:
    c256_weightnorm()
  if "cfg3" in which:
    summary("cfg3", dict(n_channels=512), T=864, wseed=7, mel_seed=2234, noise_seed=5321)
  if "cfg5" in which:
    summary("cfg5", dict(), T=4000, wseed=0, mel_seed=3234, noise_seed=6321)
