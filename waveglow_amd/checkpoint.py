"""Checkpoint dict format of the reference (src/waveglow/checkpoint.py:13-45, model_checkpoint.py:10-25):
``torch.save({state_dict, optimizer, learning_rate, iteration, hparams})``.  Field names are the format."""
from __future__ import annotations

from dataclasses import asdict, dataclass
from logging import getLogger
from pathlib import Path

import torch

from .hparams import HParams, hparams_from_dict


@dataclass
class CheckpointWaveglow:
  # Renaming any of these fields breaks existing checkpoints (checkpoint.py:14).
  state_dict: dict
  optimizer: dict
  learning_rate: float
  iteration: int
  hparams: dict

  def get_hparams(self) -> HParams:
    """hparams stored in the checkpoint; unknown keys are ignored with a warning (checkpoint.py:22-28)."""
    hp = hparams_from_dict(self.hparams)
    known = set(asdict(hp))
    ignored = {k for k in self.hparams if k not in known}
    if ignored:
      getLogger(__name__).warning(
        f"Ignored these hparams from checkpoint because they did not exist in the current HParams: {ignored}.")
    return hp

  def save(self, checkpoint_path: Path) -> None:
    getLogger(__name__).info(f"Saving model at iteration {self.iteration}...")
    torch.save(asdict(self), checkpoint_path)

  @classmethod
  def load(cls, checkpoint_path: Path, device: torch.device) -> "CheckpointWaveglow":
    checkpoint_path = Path(checkpoint_path)
    assert checkpoint_path.is_file()
    # plain tensors/dicts/scalars only: the safe loader is enough for this format
    d = torch.load(checkpoint_path, map_location=device, weights_only=True)
    return cls(**d)

  @classmethod
  def from_instances(cls, model, optimizer, hparams: HParams, iteration: int) -> "CheckpointWaveglow":
    return cls(state_dict=model.state_dict(), optimizer=optimizer.state_dict() if optimizer is not None else {},
               learning_rate=hparams.learning_rate, iteration=iteration, hparams=asdict(hparams))
