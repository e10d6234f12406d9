"""``waveglow-cli synthesize`` with the reference's flags and file conventions
(src/waveglow_cli/inference_v2.py:53-130, helper.py:8-22, defaults.py:9-10), plus real batching (the reference's
commented-out ``--batch-size``, inference_v2.py:64) and per-rank sharding of the file list under torch.distributed.run.

  python -m waveglow_amd.cli synthesize CHECKPOINT FOLDER [--sigma S] [--denoiser-strength D] [--device cuda:0]
         [--custom-hparams a=1,b=2] [--custom-seed N] [-out DIR] [-o]
  python -m waveglow_amd.cli synthesize-wav CHECKPOINT FOLDER [same flags]   wav -> mel (HIP front-end) -> wav
         (src/waveglow_cli/inference_wav.py:74-130; copy synthesis)
  python -m waveglow_amd.cli train TRAIN-FOLDER VAL-FOLDER CHECKPOINTS-FOLDER [--device cuda:0] [--custom-hparams ...]
         [--pre-trained-model CKPT --warm-start]                     (src/waveglow_cli/training.py:24-79)
  python -m waveglow_amd.cli continue-train TRAIN-FOLDER VAL-FOLDER CHECKPOINTS-FOLDER [...]   (training.py:82-124)
Under ``python -m torch.distributed.run`` the training commands run data-parallel (one process per GPU, RCCL).
"""
from __future__ import annotations

import argparse
import os
import random
import sys
from logging import getLogger
from pathlib import Path

import numpy as np
import torch

from .audio import float_to_wav, normalize_wav
from .checkpoint import CheckpointWaveglow
from .hparams import split_hparams_string
from .sharding import shard_list
from .synthesizer import Synthesizer


def _unit_float(v: str) -> float:
  f = float(v)
  if not 0 <= f <= 1:
    raise argparse.ArgumentTypeError("Value needs to be in interval [0, 1]!")
  return f


def build_parser() -> argparse.ArgumentParser:
  p = argparse.ArgumentParser(prog="waveglow-cli")
  sub = p.add_subparsers(dest="command", required=True)
  for name, desc in (("synthesize", "Synthesize mel-spectrograms to audio files (.wav)."),
                     ("synthesize-wav", "Re-synthesize audio files: wav -> mel-spectrogram -> wav.")):
    s = sub.add_parser(name, description=desc)
    s.add_argument("checkpoint", type=Path, metavar="CHECKPOINT")
    s.add_argument("folder", type=Path, metavar="FOLDER")
    s.add_argument("--sigma", type=_unit_float, default=1.0)
    s.add_argument("--denoiser-strength", type=_unit_float, default=0.0005)
    s.add_argument("--device", type=str, default="cuda:0")
    s.add_argument("--custom-hparams", type=str, default=None)
    s.add_argument("--custom-seed", type=int, default=None)
    s.add_argument("-out", "--output-directory", type=Path, default=None)
    s.add_argument("-o", "--overwrite", action="store_true")
    s.add_argument("--batch-size", type=int, default=1,
                   help="utterances per launch sequence (ragged batch; results equal one-by-one synthesis)")
  for name, desc in (("train", "Start training of a new model."), ("continue-train", "Continue training from the last checkpoint.")):
    t = sub.add_parser(name, description=desc)
    t.add_argument("train_folder", type=Path, metavar="TRAIN-FOLDER")
    t.add_argument("val_folder", type=Path, metavar="VAL-FOLDER")
    t.add_argument("checkpoints_dir", type=Path, metavar="CHECKPOINTS-FOLDER")
    t.add_argument("--device", type=str, default="cuda:0")
    t.add_argument("--custom-hparams", type=str, default=None)
    if name == "train":
      t.add_argument("--pre-trained-model", type=Path, default=None)
      t.add_argument("--warm-start", action="store_true")
  return p


def train_cmd(ns, resume: bool) -> bool:
  from .training import get_last_checkpoint, load_dataset, train
  rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
  device = torch.device(ns.device if world == 1 else f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}")
  torch.cuda.set_device(device)
  if world > 1 and not torch.distributed.is_initialized():
    torch.distributed.init_process_group("nccl", device_id=device)
  for d in (ns.train_folder, ns.val_folder):
    if not d.is_dir():
      getLogger(__name__).error(f"{d} is not a directory!")
      return False
  checkpoint = warm = None
  if resume:
    checkpoint = CheckpointWaveglow.load(get_last_checkpoint(ns.checkpoints_dir)[0], device)
  elif ns.pre_trained_model is not None and ns.warm_start:
    warm = CheckpointWaveglow.load(ns.pre_trained_model, device)
  train(custom_hparams=split_hparams_string(ns.custom_hparams), logdir=None, trainset=load_dataset(ns.train_folder),
        valset=load_dataset(ns.val_folder), save_checkpoint_dir=ns.checkpoints_dir, checkpoint=checkpoint,
        warm_model=warm, device=device)
  if world > 1:
    torch.distributed.destroy_process_group()
  return True


def synthesize(ns, from_wav: bool = False) -> bool:
  logger = getLogger(__name__)
  rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
  device = torch.device(ns.device if world == 1 else f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}")
  out_dir = ns.output_directory or ns.folder
  if out_dir.is_file():
    logger.error("Output directory is a file!")
    return False
  seed = ns.custom_seed if ns.custom_seed is not None else random.randint(1, 9999)
  try:
    ckpt = CheckpointWaveglow.load(ns.checkpoint, device)
  except Exception:
    logger.error("Checkpoint couldn't be loaded!")
    return False
  suffix = ".wav" if from_wav else ".npy"
  mel_files = sorted(p for p in ns.folder.rglob("*") if p.is_file() and p.suffix.lower() == suffix)
  mel_files = shard_list(mel_files, rank, world)
  synth = Synthesizer(ckpt, custom_hparams=split_hparams_string(ns.custom_hparams), device=device)
  taco_stft = None
  if from_wav:
    from .taco_stft import TacotronSTFT
    taco_stft = TacotronSTFT(synth.hparams, device)                   # inference_wav.py:110
  todo = []
  for mel_path in mel_files:
    wav_path = out_dir / mel_path.relative_to(ns.folder).parent / f"{mel_path.stem}.wav"
    if wav_path.exists() and not ns.overwrite:
      continue
    todo.append((mel_path, wav_path))
  bs = max(1, ns.batch_size)
  for i in range(0, len(todo), bs):
    chunk = todo[i:i + bs]
    mels = []
    for mel_path, _ in chunk:
      if from_wav:
        mels.append(taco_stft.get_mel_tensor_from_file(mel_path).unsqueeze(0))
      else:
        mels.append(torch.FloatTensor(np.load(mel_path)).unsqueeze(0))
    if len(mels) == 1:
      results = [synth.infer(mels[0], sigma=ns.sigma, denoiser_strength=ns.denoiser_strength, seed=seed)]
    else:
      results = synth.infer_batch(mels, sigma=ns.sigma, denoiser_strength=ns.denoiser_strength, seed=seed)
    for (_, wav_path), res in zip(chunk, results):
      wav_path.parent.mkdir(parents=True, exist_ok=True)
      float_to_wav(normalize_wav(res.wav_denoised), wav_path, sample_rate=res.sampling_rate)
  return True


def main(argv=None) -> int:
  ns = build_parser().parse_args(argv)
  if ns.command in ("synthesize", "synthesize-wav"):
    ok = synthesize(ns, from_wav=ns.command == "synthesize-wav")
  else:
    ok = train_cmd(ns, resume=ns.command == "continue-train")
  return 0 if ok else 1


if __name__ == "__main__":
  sys.exit(main())
