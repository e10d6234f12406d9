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
