"""``Synthesizer`` facade with the reference's surface (src/waveglow/synthesizer.py:20-94)."""
from __future__ import annotations

import datetime
import os
import random
import time
from dataclasses import dataclass
from logging import getLogger
from typing import Dict, Optional

import numpy as np
import torch

from .audio import is_overamp
from .checkpoint import CheckpointWaveglow
from .denoiser import Denoiser
from .hparams import overwrite_custom_hparams
from .model import WaveGlow


def get_default_device() -> torch.device:
  """src/waveglow/utils.py:112-118."""
  n = torch.cuda.device_count()
  if n == 1:
    return torch.device("cuda")
  if n > 1:
    return torch.device("cuda:0")
  return torch.device("cpu")


def init_global_seeds(seed: int) -> None:
  """src/waveglow/utils.py:221-229 -- called on EVERY Synthesizer.infer."""
  os.environ["PYTHONHASHSEED"] = str(seed)
  random.seed(seed)
  np.random.seed(seed)
  torch.random.manual_seed(seed)
  torch.manual_seed(seed)
  if torch.cuda.is_available():
    torch.cuda.manual_seed(seed)


@dataclass
class InferenceResult:
  wav: np.ndarray
  wav_denoised: np.ndarray
  sampling_rate: int
  inference_duration_s: float
  denoising_duration_s: float
  was_overamplified: bool
  timepoint: datetime.datetime


def load_model(hparams, state_dict: Optional[dict], device: torch.device) -> WaveGlow:
  """src/waveglow/train.py:48-55 (without the silent CPU fallback of try_copy_to: the kernels need the GPU)."""
  model = WaveGlow(hparams).to(device)
  if state_dict is not None:
    model.load_state_dict(state_dict)
  return model


class Synthesizer:
  def __init__(self, checkpoint: CheckpointWaveglow, *, custom_hparams: Optional[Dict[str, str]] = None,
               device: Optional[torch.device] = None):
    device = torch.device(device) if device is not None else get_default_device()
    if device.type == "cuda" and device.index is None:
      device = torch.device("cuda", torch.cuda.current_device())
    hparams = overwrite_custom_hparams(checkpoint.get_hparams(), custom_hparams)
    model = load_model(hparams, checkpoint.state_dict, device)
    model = WaveGlow.remove_weightnorm(model).eval()
    self.device, self.hparams, self.model = device, hparams, model
    self.denoiser = Denoiser(waveglow=model, hparams=hparams, mode="zeros", device=device).to(device)

  def infer(self, mel: torch.Tensor, *, sigma: float = 1.0, denoiser_strength: float = 0.0005,
            seed: int = 0) -> InferenceResult:
    timepoint = datetime.datetime.now()
    init_global_seeds(seed)
    denoising_duration = 0
    mel = mel.to(self.device)
    start = time.perf_counter()
    with torch.no_grad():
      audio = self.model.infer(mel, sigma=sigma)
      torch.cuda.synchronize(self.device)      # the reference's timer has no device sync (synthesizer.py:61)
      end = time.perf_counter()
      audio_denoised = audio
      if denoiser_strength > 0:
        t0 = time.perf_counter()
        audio_denoised = self.denoiser(audio, strength=denoiser_strength)
        torch.cuda.synchronize(self.device)
        denoising_duration = time.perf_counter() - t0
    audio_np = audio.squeeze().float().cpu().numpy()
    audio_denoised_np = audio_denoised.squeeze().float().cpu().numpy()
    over = bool(is_overamp(audio_np))
    if over:
      getLogger(__name__).debug("Waveglow output was overamplified.")
    return InferenceResult(wav=audio_np, wav_denoised=audio_denoised_np, sampling_rate=self.hparams.sampling_rate,
                           inference_duration_s=end - start, denoising_duration_s=denoising_duration,
                           was_overamplified=over, timepoint=timepoint)

  def infer_batch(self, mels, *, sigma: float = 1.0, denoiser_strength: float = 0.0005, seed: int = 0):
    """Several mel-spectrograms ``[1 or -, n_mel, T_i]`` of different lengths in ONE ragged launch sequence (the
    reference's commented-out ``--batch-size``, inference_v2.py:64).  Every utterance gets the audio that ``infer`` on
    it alone would return with the same seed: its noise is drawn exactly as that call draws it (seed reset, then the
    three tensors in the reference's order and shapes, model.py:234-271), and the kernels treat the padding behind an
    utterance as the end of the sequence (``wg_infer_ragged``).  Returns one InferenceResult per utterance; the
    durations are the batch's, divided evenly."""
    timepoint = datetime.datetime.now()
    mels = [m.squeeze(0) if m.dim() == 3 else m for m in mels]
    B = len(mels)
    lens = [int(m.shape[1]) for m in mels]
    Tm = max(lens)
    model, dev = self.model, self.device
    n_mel, ng = mels[0].shape[0], model.n_group
    dtype = mels[0].dtype
    mel = torch.zeros((B, n_mel, Tm), dtype=dtype, device=dev)
    z_init = torch.zeros((B, model.n_remaining_channels, Tm * 256 // ng), dtype=dtype, device=dev)
    early = [k for k in reversed(range(model.n_flows)) if k % model.n_early_every == 0 and k > 0]
    z_early = [torch.zeros((B, model.n_early_size, Tm * 256 // ng), dtype=dtype, device=dev) for _ in early]
    for b, m in enumerate(mels):
      L = lens[b] * 256 // ng
      mel[b, :, :lens[b]] = m.to(dev)
      init_global_seeds(seed)                                                        # synthesizer.py:56
      z_init[b, :, :L] = torch.empty((1, model.n_remaining_channels, L), dtype=dtype, device=dev).normal_()[0]
      for j in range(len(early)):
        z_early[j][b, :, :L] = torch.empty((1, model.n_early_size, L), dtype=dtype, device=dev).normal_()[0]
    start = time.perf_counter()
    with torch.no_grad():
      audio = model.infer_with_noise(mel, z_init, z_early, sigma, frames=torch.tensor(lens, dtype=torch.int32))
      torch.cuda.synchronize(dev)
      end = time.perf_counter()
      t0 = time.perf_counter()
      outs = []
      for b in range(B):
        a = audio[b:b + 1, :256 * lens[b]]
        d = self.denoiser(a.contiguous(), strength=denoiser_strength) if denoiser_strength > 0 else a
        outs.append((a, d))
      torch.cuda.synchronize(dev)
      den = time.perf_counter() - t0 if denoiser_strength > 0 else 0
    res = []
    for a, d in outs:
      a_np, d_np = a.squeeze().float().cpu().numpy(), d.squeeze().float().cpu().numpy()
      res.append(InferenceResult(wav=a_np, wav_denoised=d_np, sampling_rate=self.hparams.sampling_rate,
                                 inference_duration_s=(end - start) / B, denoising_duration_s=den / B,
                                 was_overamplified=bool(is_overamp(a_np)), timepoint=timepoint))
    return res
