"""Training loop with the reference's semantics (src/waveglow/train.py:93-238, dataloader.py:16-104,
utils.py:361-470): same ``train(...)`` signature, hparams handling, data pipeline (random segment -> mel on the device),
Adam on the weight-normed parameters, iteration / epoch bookkeeping, resume and checkpoint cadence, validation loss.
Every step's compute -- mel front-end, forward, loss, backward -- runs in the HIP library; with ``torch.distributed``
initialised the file list is sharded by rank and the gradients are averaged (waveglow_amd/distributed.py).
Not reproduced: tensorboard logging (``WaveglowLogger``) and the validation plots / MCD metrics.
"""
from __future__ import annotations

import random
import time
from dataclasses import dataclass
from logging import getLogger
from math import floor
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from .audio import get_wav_tensor_segment
from .checkpoint import CheckpointWaveglow
from .distributed import GradientAllReducer
from .hparams import HParams, overwrite_custom_hparams
from .model import WaveGlow, WaveGlowLoss
from .sharding import equal_shard_list
from .taco_stft import TacotronSTFT

PYTORCH_EXT = ".pt"


@dataclass
class Entry:
  """src/waveglow/typing.py"""
  stem: str
  basename: str
  wav_absolute_path: Path


def load_dataset(folder: Path) -> List[Entry]:
  """All ``*.wav`` files below ``folder`` (waveglow_cli/parser.py: load_dataset)."""
  folder = Path(folder)
  return [Entry(p.stem, p.name, p.absolute()) for p in sorted(folder.rglob("*.wav"))]


# ---------------------------------------------------------------- iteration bookkeeping (utils.py:361-470)
@dataclass
class SaveIterationSettings:
  epochs: int
  batch_iterations: int
  save_first_iteration: bool
  save_last_iteration: bool
  iters_per_checkpoint: int
  epochs_per_checkpoint: int


def iteration_to_epoch(iteration: int, batch_iterations: int) -> int:
  assert iteration > 0
  return floor((iteration - 1) / batch_iterations)


def iteration_to_batch_iteration(iteration: int, batch_iterations: int) -> int:
  assert iteration > 0
  return (iteration - 1) % batch_iterations


def get_continue_epoch(current_iteration: int, batch_iterations: int) -> int:
  return iteration_to_epoch(current_iteration + 1, batch_iterations)


def get_continue_batch_iteration(iteration: int, batch_iterations: int) -> int:
  return iteration_to_batch_iteration(iteration + 1, batch_iterations)


def skip_batch(continue_batch_iteration: int, batch_iteration: int) -> bool:
  return batch_iteration < continue_batch_iteration


def check_is_first(iteration: int) -> bool:
  assert iteration >= 0
  return iteration == 1


def check_is_last(iteration: int, epochs: int, batch_iterations: int) -> bool:
  assert iteration >= 0
  return iteration == epochs * batch_iterations


def check_is_save_iteration(iteration: int, iters_per_checkpoint: int) -> bool:
  assert iteration >= 0
  return iteration > 0 and iters_per_checkpoint > 0 and iteration % iters_per_checkpoint == 0


def check_is_save_epoch(epoch: int, epochs_per_checkpoint: int) -> bool:
  assert epoch >= 0
  return epochs_per_checkpoint > 0 and ((epoch + 1) % epochs_per_checkpoint == 0)


def check_is_last_batch_iteration(iteration: int, batch_iterations: int) -> bool:
  assert iteration >= 0 and batch_iterations > 0
  if iteration == 0:
    return False
  return iteration_to_batch_iteration(iteration, batch_iterations) + 1 == batch_iterations


def check_save_it(epoch: int, iteration: int, settings: SaveIterationSettings) -> bool:
  if check_is_first(iteration) and settings.save_first_iteration:
    return True
  if check_is_last(iteration, settings.epochs, settings.batch_iterations) and settings.save_last_iteration:
    return True
  if check_is_save_iteration(iteration, settings.iters_per_checkpoint):
    return True
  return (check_is_last_batch_iteration(iteration, settings.batch_iterations)
          and check_is_save_epoch(epoch, settings.epochs_per_checkpoint))


def get_pytorch_filename(name) -> str:
  return f"{name}{PYTORCH_EXT}"


def get_all_checkpoint_iterations(checkpoint_dir: Path) -> List[int]:
  return sorted(int(p.name[:-len(PYTORCH_EXT)]) for p in Path(checkpoint_dir).iterdir()
                if p.is_file() and p.name.endswith(PYTORCH_EXT) and p.name[:-len(PYTORCH_EXT)].isdigit())


def get_last_checkpoint(checkpoint_dir: Path) -> Tuple[Path, int]:
  its = get_all_checkpoint_iterations(checkpoint_dir)
  if not its:
    raise Exception("No checkpoint iteration found!")
  return Path(checkpoint_dir) / get_pytorch_filename(max(its)), max(its)


# ---------------------------------------------------------------- data (dataloader.py:16-104)
class MelLoader(Dataset):
  """Random training segment of every wav + its mel spectrogram, both on the device (dataloader.py:16-57)."""

  def __init__(self, entries: List[Entry], hparams: HParams, device: torch.device):
    self.device = torch.device(device)
    self.taco_stft = TacotronSTFT(hparams, self.device)
    self.hparams = hparams
    data = list(entries)
    random.seed(hparams.seed)
    random.shuffle(data)
    self.wav_paths = {i: e.wav_absolute_path for i, e in enumerate(data)}
    self.cache = None
    if hparams.cache_wavs:
      self.cache = {i: self.taco_stft.get_wav_tensor_from_file(p) for i, p in self.wav_paths.items()}

  def __getitem__(self, index):
    wav = (self.cache[index].clone() if self.cache is not None
           else self.taco_stft.get_wav_tensor_from_file(self.wav_paths[index]))
    wav = get_wav_tensor_segment(wav, self.hparams.segment_length).to(self.device)
    return self.taco_stft.get_mel_tensor(wav), wav

  def __len__(self):
    return len(self.wav_paths)


def parse_batch(batch):
  mel, audio = batch
  return (mel, audio), (mel, audio)


def prepare_trainloader(hparams: HParams, trainset: List[Entry], device) -> DataLoader:
  return DataLoader(MelLoader(trainset, hparams, device), num_workers=0, shuffle=False, sampler=None,
                    batch_size=hparams.batch_size, pin_memory=False, drop_last=True)


def prepare_valloader(hparams: HParams, valset: List[Entry], device) -> DataLoader:
  return DataLoader(MelLoader(valset, hparams, device), num_workers=0, shuffle=False, sampler=None,
                    batch_size=hparams.batch_size, pin_memory=False)


# ---------------------------------------------------------------- model / optimiser (train.py:48-55, :241-266)
def load_model(hparams: HParams, state_dict: Optional[dict], device) -> WaveGlow:
  model = WaveGlow(hparams)
  if state_dict is not None:
    model.load_state_dict(state_dict)
  return model.to(device)


def load_optimizer(model_parameters, hparams: HParams, state_dict: Optional[dict]) -> torch.optim.Adam:
  # same optimiser and hyper-parameters as the reference (train.py:58-66); `fused` is torch's single-kernel implementation
  # of the same update (0.8 ms instead of 2.2 ms per step for the 686 parameter tensors at 256 channels)
  params = list(model_parameters)
  fused = len(params) > 0 and all(p.is_cuda for p in params)
  optimizer = torch.optim.Adam(params=params, lr=hparams.learning_rate, fused=fused)
  if state_dict is not None:
    optimizer.load_state_dict(state_dict)
  return optimizer


def validate_model(model, criterion, val_loader) -> float:
  """Average validation loss (utils.py:330-357); runs the no-grad forward of the library."""
  model.eval()
  losses = []
  with torch.no_grad():
    for batch in val_loader:
      x, y = parse_batch(batch)
      losses.append(float(criterion(model(x), y)))
  model.train()
  return float(np.mean(losses)) if losses else float("nan")


def train(custom_hparams: Optional[Dict[str, str]], logdir: Optional[Path], trainset: List[Entry], valset: List[Entry],
          save_checkpoint_dir: Path, checkpoint: Optional[CheckpointWaveglow], warm_model: Optional[CheckpointWaveglow],
          device: torch.device, max_iterations: Optional[int] = None) -> List[float]:
  """train.py:93-238.  ``max_iterations`` (not in the reference) stops after that many optimiser steps.
  Returns the training losses of the executed steps."""
  logger = getLogger(__name__)
  complete_start = time.time()
  device = torch.device(device)
  hparams = checkpoint.get_hparams() if checkpoint is not None else HParams()
  hparams = overwrite_custom_hparams(hparams, custom_hparams)
  torch.manual_seed(hparams.seed)                                     # init_torch (train.py:79-82)
  torch.cuda.manual_seed(hparams.seed)

  dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
  rank = torch.distributed.get_rank() if dist_on else 0
  world = torch.distributed.get_world_size() if dist_on else 1
  if world > 1:
    # equal shards: every rank takes the same number of steps per epoch (the all-reduces inside backward() pair up
    # step by step); up to world-1 files at the end of the list sit out
    trainset = equal_shard_list(list(trainset), rank, world)

  model = load_model(hparams, checkpoint.state_dict if checkpoint is not None else None, device)
  optimizer = load_optimizer(model.parameters(), hparams, checkpoint.optimizer if checkpoint is not None else None)
  iteration = checkpoint.iteration if checkpoint is not None else 0
  if checkpoint is None and warm_model is not None:
    logger.info("Loading states from pretrained model...")
    model.load_state_dict(warm_model.state_dict)                      # warm_start_model (train.py:85-90)
  criterion = WaveGlowLoss(sigma=hparams.sigma)
  # data parallel: gradients are averaged inside backward(), flow by flow, overlapped with the rest of the backward
  # pass (waveglow_amd/train.py); GradientAllReducer is the unfused alternative for callers that prefer it
  from .train import enable_data_parallel
  reducer = None
  if world > 1 and not enable_data_parallel(model):
    reducer = GradientAllReducer(model.parameters())

  train_loader = prepare_trainloader(hparams, trainset, device)
  val_loader = prepare_valloader(hparams, valset, device)
  batch_iterations = len(train_loader)
  if batch_iterations == 0:
    logger.error("Not enough training data.")
    raise Exception()
  if world > 1:
    n = torch.tensor([batch_iterations, -batch_iterations], dtype=torch.int64,
                     device=device if torch.distributed.get_backend() == "nccl" else "cpu")
    torch.distributed.all_reduce(n, op=torch.distributed.ReduceOp.MAX)
    if int(n[0]) != batch_iterations or int(-n[1]) != batch_iterations:
      raise Exception(f"rank {rank}: {batch_iterations} steps per epoch, other ranks between {int(-n[1])} and {int(n[0])}")

  model.train()
  train_start = time.perf_counter()
  start = train_start
  settings = SaveIterationSettings(epochs=hparams.epochs, batch_iterations=batch_iterations, save_first_iteration=True,
                                   save_last_iteration=True, iters_per_checkpoint=hparams.iters_per_checkpoint,
                                   epochs_per_checkpoint=hparams.epochs_per_checkpoint)
  losses: List[float] = []
  batch_durations: List[float] = []
  continue_epoch = get_continue_epoch(iteration, batch_iterations)
  for epoch in range(continue_epoch, hparams.epochs):
    next_batch_iteration = get_continue_batch_iteration(iteration, batch_iterations)
    for batch_iteration, batch in enumerate(train_loader):
      if skip_batch(batch_iteration=batch_iteration, continue_batch_iteration=next_batch_iteration):
        continue
      model.zero_grad()
      x, y = parse_batch(batch)
      loss = criterion(model(x), y)
      loss.backward()
      if reducer is not None:
        reducer.reduce()
      # ONE host sync per step, after everything has been queued.  The reference reads the loss before backward()
      # (train.py:193-195) -- same value here, without a GPU bubble while the backward pass's ~700 launches are issued; and
      # with the fused optimiser the overflow check gates the update ON THE DEVICE (torch's found_inf hook, as
      # GradScaler uses it: the step is skipped, nothing is written), so the host reads the flag only afterwards.
      finite = getattr(model, "grad_finite", None)      # set by the library's backward (waveglow_amd/train.py)
      if reducer is not None and finite is not None:
        # the unfused exchange ran AFTER backward() computed the flag, so the flag is rank-local while the summed gradients
        # are not: every rank must take the same branch below (the per-flow path computes its flag after the all-reduce)
        bad = (~finite).to(torch.float32).reshape(1)
        torch.distributed.all_reduce(bad, op=torch.distributed.ReduceOp.MAX, group=reducer.group)
        finite = bad.reshape(()) == 0
        model.grad_finite = finite
      gated = finite is not None and any(g.get("fused") for g in optimizer.param_groups)
      if gated:
        optimizer.found_inf = (~finite).to(torch.float32).reshape(())
      elif finite is not None and not bool(finite):
        from .train import nonfinite_message
        raise Exception(nonfinite_message(float(getattr(model, "grad_scale", 0.0))))
      optimizer.step()
      reduced_loss = loss.item()
      if gated and not bool(finite):
        from .train import nonfinite_message
        raise Exception(nonfinite_message(float(getattr(model, "grad_scale", 0.0))))
      iteration += 1
      losses.append(reduced_loss)
      end = time.perf_counter()
      batch_durations.append(end - start)
      start = end
      logger.info(" | ".join([
        f"Epoch: {epoch + 1}/{hparams.epochs}", f"Iteration: {batch_iteration + 1}/{batch_iterations}",
        f"Total iteration: {iteration}/{hparams.epochs * batch_iterations}", f"Train loss: {reduced_loss:.6f}",
        f"Duration: {batch_durations[-1]:.2f}s/it", f"Avg. duration: {np.mean(batch_durations):.2f}s/it",
        f"Total Duration: {(time.perf_counter() - train_start) / 60 / 60:.2f}h"]))
      if check_save_it(epoch, iteration, settings):
        if rank == 0:
          ckpt = CheckpointWaveglow.from_instances(model=model, optimizer=optimizer, hparams=hparams, iteration=iteration)
          Path(save_checkpoint_dir).mkdir(parents=True, exist_ok=True)
          ckpt.save(Path(save_checkpoint_dir) / get_pytorch_filename(iteration))
        logger.info(f"Validation loss {iteration}: {validate_model(model, criterion, val_loader):9f}")
      if max_iterations is not None and len(losses) >= max_iterations:
        return losses
    # a resumed epoch starts in the middle; the following ones from batch 0
  logger.info(f"Finished training. Total duration: {(time.time() - complete_start) / 60:.2f}m")
  return losses
