"""The few wav helpers the synthesize path needs (src/waveglow/audio_utils.py:26-32, :36-64, :67-95, :132-138)."""
from __future__ import annotations

import numpy as np
from scipy.io.wavfile import write


def _min_max(dtype):
  if dtype == np.int16:
    return -32768, 32767
  if dtype == np.int32:
    return -2147483648, 2147483647
  if dtype in (np.float32, np.float64):
    return -1.0, 1.0
  raise AssertionError(dtype)


def is_overamp(wav: np.ndarray) -> bool:
  lo, hi = _min_max(wav.dtype)
  return bool(np.min(wav) < lo or np.max(wav) > hi)


def normalize_wav(wav: np.ndarray) -> np.ndarray:
  """Scale so that max |x| hits the dtype's maximum (audio_utils.py:67-95)."""
  lo, hi = _min_max(wav.dtype)
  if wav.dtype in (np.int16, np.int32) and np.min(wav) == lo:
    return wav
  max_val = np.max(np.abs(wav))
  if max_val != hi and max_val != 0:
    orig = wav.dtype
    wf = wav.astype(np.float32) * hi / max_val
    if orig in (np.int16, np.int32):
      wf = np.round(wf, 0)
    wav = wf.astype(orig)
  # the reference checks its own result (audio_utils.py:92-93) -- which makes it REJECT inputs whose float32 round trip
  # does not land exactly on the maximum (float64 signals, large int32): same behaviour here
  assert np.max(np.abs(wav)) == hi or np.max(np.abs(wav)) == 0
  assert not is_overamp(wav)
  return wav


def convert_wav(wav: np.ndarray, to_dtype) -> np.ndarray:
  """float [-1,1] -> integer PCM with rounding (audio_utils.py:36-64)."""
  if wav.dtype != to_dtype:
    _, hi = _min_max(to_dtype)
    cur_lo, _ = _min_max(wav.dtype)
    wav = wav / (-1 * cur_lo) * hi              # the reference divides by -min of the source type
    if to_dtype in (np.int16, np.int32):
      wav = np.round(wav, 0)
    wav = wav.astype(to_dtype)
  return wav


def float_to_wav(wav: np.ndarray, path, dtype=np.int16, sample_rate: int = 22050) -> None:
  write(filename=path, rate=sample_rate, data=convert_wav(wav, dtype))


def wav_to_float32(path):
  """(float32 samples in [-1, 1], sampling rate) of a PCM wav file (audio_utils.py:206-216)."""
  from scipy.io.wavfile import read
  sampling_rate, wav = read(path)
  return convert_wav(wav, np.float32), sampling_rate


def get_wav_tensor_segment(wav_tensor, segment_length: int):
  """Random segment of the training length, or zero padding up to it (audio_utils.py:141-150)."""
  import random
  import torch
  if wav_tensor.size(0) >= segment_length:
    audio_start = random.randint(0, wav_tensor.size(0) - segment_length)
    return wav_tensor[audio_start:audio_start + segment_length]
  return torch.nn.functional.pad(wav_tensor, (0, segment_length - wav_tensor.size(0)), "constant").data
