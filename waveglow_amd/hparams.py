"""Hyper-parameters with the reference's field names (they are stored by name in
checkpoints: src/waveglow/model_checkpoint.py:23, src/waveglow/hparams.py:6-43,
src/waveglow/taco_stft.py:36-50).  Re-declared, not imported."""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass
from typing import Dict, Optional


@dataclass
class HParams:
  # OptimizerHParams (hparams.py:34-38)
  learning_rate: float = 1e-4
  sigma: float = 1.0
  batch_size: int = 1
  # ModelHParams (hparams.py:19-31)
  segment_length: int = 16000
  n_mel_channels: int = 80
  n_flows: int = 12
  n_group: int = 8
  n_early_every: int = 4
  n_early_size: int = 2
  n_layers: int = 8
  n_channels: int = 256
  kernel_size: int = 3
  # TSTFTHParams / STFTHParams (taco_stft.py:36-50)
  filter_length: int = 1024
  hop_length: int = 256
  win_length: int = 1024
  window: str = "hann"
  sampling_rate: int = 22050
  mel_fmin: float = 0.0
  mel_fmax: float = 8000.0
  # ExperimentHParams (hparams.py:6-16)
  epochs: int = 100000
  iters_per_checkpoint: int = 2000
  epochs_per_checkpoint: int = 1
  seed: int = 1234
  cache_wavs: bool = False
  cudnn_enabled: bool = True
  cudnn_benchmark: bool = False


def split_hparams_string(hparams: Optional[str]) -> Optional[Dict[str, str]]:
  """``"a=1,b=2"`` -> dict (src/waveglow/utils.py:32-38)."""
  if hparams is None:
    return None
  return dict([x.split("=") for x in hparams.split(",")])


def overwrite_custom_hparams(hp: HParams, custom: Optional[Dict[str, str]]) -> HParams:
  """Typed override by the default's type; unknown key -> bare ``Exception``
  (src/waveglow/utils.py:48-59, :62-97)."""
  if custom is None:
    return hp
  fields = {f.name for f in dataclasses.fields(hp)}
  conv = {}
  for key, raw in custom.items():
    if key not in fields:
      raise Exception()
  for key, raw in custom.items():
    # utils.py:76-82: "" -> None, otherwise type(default)(value)
    conv[key] = None if raw == "" else type(getattr(hp, key))(raw)
  return dataclasses.replace(hp, **conv)


def hparams_from_dict(d: Dict) -> HParams:
  """Checkpoint dict -> HParams, dropping unknown keys (src/waveglow/checkpoint.py:22-28)."""
  names = {f.name for f in dataclasses.fields(HParams)}
  return HParams(**{k: v for k, v in d.items() if k in names})
