"""Generate golden vectors by running the REFERENCE's own model code (build container only).

  python tests/golden/make_golden.py

Imports /root/reference/src/waveglow/model.py (via _ref_import), loads the deterministic
synthetic weights of waveglow_amd.synthetic into the reference modules, runs
``WaveGlow.infer`` (model.py:223-274, noise replayed from the global CPU RNG in the
reference's draw order), ``WaveGlow.forward`` (model.py:178-221) and ``WaveGlowLoss``
(train.py:31-45), and writes inputs + expected outputs as .npz fixtures next to this
script.  Fixtures are data only; no reference source is copied.
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

CASES = {
  # name: (hparam overrides, B, T, weight seed, sigma)
  "tiny": (dict(n_channels=16, n_layers=3, n_flows=4, n_early_every=2), 2, 6, 3, 0.8),
  "c64": (dict(n_channels=64, n_layers=4, n_flows=6, n_early_every=2), 2, 12, 5, 0.6),
  "c256": (dict(), 2, 8, 0, 0.6),
  "c512": (dict(n_channels=512), 1, 4, 7, 0.6),
}


def build_reference(hp_over, sd, normed=False):
  hp = ref_hparams.HParams(**hp_over)
  model = ref_model.WaveGlow(hp)
  if normed:
    model.load_state_dict(synthetic.to_weightnorm_form(sd))
    model = ref_model.WaveGlow.remove_weightnorm(model)
  else:
    model = ref_model.WaveGlow.remove_weightnorm(model)
    model.load_state_dict(sd)
  return model.eval()


def run_case(name, hp_over, B, T, wseed, sigma):
  hp = HParams(**hp_over)
  sd = synthetic.make_state_dict(hp, seed=wseed)
  model = build_reference(hp_over, sd)
  mel = synthetic.make_mel(B, T, hp.n_mel_channels, seed=1234 + B + T)
  L = T * 256 // hp.n_group
  noise_seed = 4321 + T
  out = {}
  with torch.no_grad():
    # ---- infer: seed the global CPU RNG, let the reference draw, then replay the draws
    torch.manual_seed(noise_seed)
    audio = model.infer(mel, sigma=sigma)
    torch.manual_seed(noise_seed)
    n_rem = synthetic.flow_channels(hp)[-1]
    z_init = torch.FloatTensor(B, n_rem, L).normal_()
    for k in reversed(range(hp.n_flows)):
      if k % hp.n_early_every == 0 and k > 0:
        out[f"z_early_{k}"] = torch.FloatTensor(B, hp.n_early_size, L).normal_().numpy()
    out["mel"] = mel.numpy()
    out["z_init"] = z_init.numpy()
    out["audio"] = audio.numpy()
    # ---- forward + loss on a synthetic waveform of S = 256*T - 96 samples (exercises the crop, model.py:187-189)
    S = 256 * T - 96
    g = torch.Generator().manual_seed(99 + T)
    wav = (torch.rand(B, S, generator=g) * 0.6 - 0.3)
    z, log_s_list, log_det_list = model((mel, wav))
    # snapshot before the loss: train.py:38-42 accumulates IN PLACE into log_det_W_list[0]
    out["fwd_log_det"] = np.array([float(x) for x in log_det_list], dtype=np.float32)
    loss = ref_train.WaveGlowLoss(sigma=1.0)((z, log_s_list, log_det_list), None)
    out["fwd_audio_in"] = wav.numpy()
    out["fwd_z"] = z.numpy()
    for k, ls in enumerate(log_s_list):
      out[f"fwd_log_s_{k}"] = ls.numpy()
    out["fwd_loss"] = np.array(float(loss), dtype=np.float32)
    # ---- weight-normed checkpoint form (686-key style) through the reference's own fold
    if name in ("tiny", "c64"):
      model_n = build_reference(hp_over, sd, normed=True)
      torch.manual_seed(noise_seed)
      out["audio_from_weightnorm_ckpt"] = model_n.infer(mel, sigma=sigma).numpy()
  out["sigma"] = np.array(sigma, dtype=np.float32)
  out["weight_seed"] = np.array(wseed)
  out["hp_json"] = np.array(repr(sorted(hp_over.items())))
  # checksum of the generated weights, so a generator drift is caught before a numeric mismatch
  crc = 0
  for key in sorted(sd):
    crc = zlib.crc32(sd[key].numpy().tobytes(), crc)
  out["weights_crc32"] = np.array(crc, dtype=np.uint32)
  np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
  print(f"{name}: audio {audio.shape} rms={audio.pow(2).mean().sqrt():.4f} max={audio.abs().max():.3f} "
        f"loss={float(loss):.5f} crc={crc:#x}")


if __name__ == "__main__":
  torch.set_num_threads(8)
  for name, (hp_over, B, T, wseed, sigma) in CASES.items():
    run_case(name, hp_over, B, T, wseed, sigma)
