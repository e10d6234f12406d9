"""NVIDIA WaveGlow checkpoint -> this repository's checkpoint format (src/waveglow/converter/convert.py:37-94).

The reference unpickles NVIDIA's file (a whole pickled ``WaveGlow`` module, which executes code from the file) and takes
``model.state_dict()``.  Here only the SAFE route exists: a file that holds plain tensors -- a bare state_dict, or
``{"state_dict": ...}`` / ``{"model": state_dict}`` -- loaded with ``weights_only=True``.  NVIDIA's parameter names are the
reference's (``WN.k.in_layers.i.weight_g / weight_v`` from the old ``torch.nn.utils.weight_norm`` are accepted by
``WaveGlow.load_state_dict`` here, as are the fused ``.weight`` keys of a checkpoint saved after ``remove_weightnorm``).
A pickled module is refused with instructions instead of being executed.
"""
from __future__ import annotations

from dataclasses import asdict
from pathlib import Path

import torch

from .checkpoint import CheckpointWaveglow
from .hparams import HParams


def nvidia_hparams() -> HParams:
  """The published training configuration (convert.py:45-66: WaveGlow paper values)."""
  return HParams(sampling_rate=22050, n_mel_channels=80, filter_length=1024, hop_length=256, win_length=1024,
                 batch_size=24, learning_rate=1e-4, n_early_every=4, n_early_size=2, n_layers=8, segment_length=16000)


def convert_glow_state_dict(state_dict: dict, iteration: int = 580000) -> CheckpointWaveglow:
  """convert.py:68-92 on an already-extracted state_dict ("for 580,000 iterations")."""
  hp = nvidia_hparams()
  sd = {k: v for k, v in state_dict.items() if torch.is_tensor(v)}
  return CheckpointWaveglow(state_dict=sd, optimizer={}, learning_rate=hp.learning_rate, iteration=iteration,
                            hparams=asdict(hp))


def convert_glow(source: Path, device: torch.device = torch.device("cpu")) -> CheckpointWaveglow:
  try:
    obj = torch.load(Path(source), map_location=device, weights_only=True)
  except Exception as ex:   # pickled module objects need arbitrary code execution
    raise RuntimeError(
      f"{source} is not a plain-tensor checkpoint (NVIDIA's original files pickle the whole model object). Extract its "
      "state_dict in an environment you trust -- torch.save(torch.load(path, weights_only=False)['model'].state_dict(), "
      "out) -- and convert that file.") from ex
  for key in ("state_dict", "model"):
    if isinstance(obj, dict) and isinstance(obj.get(key), dict):
      obj = obj[key]
  if not isinstance(obj, dict) or not any(k.startswith("WN.") for k in obj):
    raise RuntimeError(f"{source} does not contain a WaveGlow state_dict")
  return convert_glow_state_dict(obj)


def convert_glow_files(origin: Path, destination: Path, device: torch.device = torch.device("cpu"),
                       keep_orig: bool = False) -> CheckpointWaveglow:
  """convert.py:19-35 (the original is kept as ``<origin>.orig`` when converting in place with keep_orig)."""
  import os
  import shutil
  origin, destination = Path(origin), Path(destination)
  res = convert_glow(origin, device)
  tmp = destination.with_suffix(destination.suffix + ".tmp")
  res.save(tmp)
  if keep_orig:
    if origin == destination:
      shutil.move(origin, Path(f"{origin.absolute()}.orig"))
  else:
    os.remove(origin)
  shutil.move(tmp, destination)
  return res
