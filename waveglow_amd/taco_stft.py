"""Mel front-end with the reference's surface (src/waveglow/taco_stft.py:53-125), computed by the HIP library
(``wg_stft_mel``: conv-STFT magnitudes on exact-fp32 MFMA, mel projection + log compression).

``librosa.filters.mel`` (taco_stft.py:66-73) is restated below (Slaney scale, Slaney area normalisation -- librosa's
defaults ``htk=False, norm='slaney'``); librosa is absent here, so this front-end is pinned only against
``oracle/stft_oracle.py`` (numpy fp64 restatement), not against reference outputs: **parity unpinned**.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from .audio import wav_to_float32
from .denoiser import stft_bases

FLOAT32_64_MIN_WAV, FLOAT32_64_MAX_WAV = -1.0, 1.0   # audio_utils.py


@dataclass
class STFTHParams:
  filter_length: int = 1024
  hop_length: int = 256
  win_length: int = 1024
  window: str = "hann"


@dataclass
class TSTFTHParams(STFTHParams):
  n_mel_channels: int = 80
  sampling_rate: int = 22050
  mel_fmin: float = 0.0
  mel_fmax: float = 8000.0


def _hz_to_mel(f):
  f = np.asarray(f, dtype=np.float64)
  f_sp = 200.0 / 3
  mels = f / f_sp
  min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
  min_log_mel = min_log_hz / f_sp
  return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
  m = np.asarray(m, dtype=np.float64)
  f_sp = 200.0 / 3
  min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
  min_log_mel = min_log_hz / f_sp
  return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_mel_filterbank(sr: int, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
  """librosa.filters.mel(sr=, n_fft=, n_mels=, fmin=, fmax=) with its defaults: triangular filters on the Slaney mel
  scale, each scaled to unit area (2 / bandwidth).  [n_mels, 1 + n_fft/2] float32."""
  fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
  mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
  fdiff = np.diff(mel_f)
  ramps = mel_f[:, None] - fftfreqs[None, :]
  lower = -ramps[:-2] / fdiff[:-1, None]
  upper = ramps[2:] / fdiff[1:, None]
  weights = np.maximum(0.0, np.minimum(lower, upper))
  weights *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
  return weights.astype(np.float32)


def dynamic_range_compression(x, C_=1, clip_val=1e-5):
  return torch.log(torch.clamp(x, min=clip_val) * C_)


def dynamic_range_decompression(x, C_=1):
  return torch.exp(x) / C_


class TacotronSTFT(torch.nn.Module):
  def __init__(self, hparams, device):
    super().__init__()
    device = torch.device(device)
    if device.type != "cuda":
      raise _lib.WgError("the mel front-end runs on the GPU library only")
    self.n_mel_channels = hparams.n_mel_channels
    self.sampling_rate = hparams.sampling_rate
    self.device = device
    self.lib = _lib.load()
    fwd, inv, wsq = stft_bases(hparams.filter_length, hparams.hop_length, hparams.win_length, hparams.window)
    self._h = C.c_void_p()
    _lib.check(self.lib.wg_stft_create(fwd.ctypes.data, inv.ctypes.data, wsq.ctypes.data, hparams.filter_length,
                                       hparams.hop_length, _lib.device_index(device), C.byref(self._h)))
    basis = slaney_mel_filterbank(hparams.sampling_rate, hparams.filter_length, hparams.n_mel_channels,
                                  hparams.mel_fmin, hparams.mel_fmax)
    self.register_buffer("mel_basis", torch.from_numpy(basis).to(device))

  def __del__(self):
    try:
      if getattr(self, "_h", None):
        self.lib.wg_stft_destroy(self._h)
    except Exception:
      pass

  def spectral_normalize(self, magnitudes):
    return dynamic_range_compression(magnitudes)

  def spectral_de_normalize(self, magnitudes):
    return dynamic_range_decompression(magnitudes)

  def mel_spectrogram(self, y: torch.Tensor) -> torch.Tensor:
    """(B, T) in [-1, 1] -> (B, n_mel_channels, T // hop + 1)   (taco_stft.py:84-104)"""
    assert float(y.min()) >= FLOAT32_64_MIN_WAV and float(y.max()) <= FLOAT32_64_MAX_WAV   # taco_stft.py:95-97
    y = y.to(self.device, torch.float32).contiguous()
    B, N = y.shape
    nbytes = self.lib.wg_stft_mel_workspace_bytes(self._h, B, N)
    if nbytes == 0:
      raise _lib.WgError(f"mel front-end: audio of {N} samples is too short (reflect padding needs > 512)")
    F_ = N // 256 + 1
    out = torch.empty((B, self.n_mel_channels, F_), dtype=torch.float32, device=self.device)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
    stream = torch.cuda.current_stream(self.device).cuda_stream
    _lib.check(self.lib.wg_stft_mel(self._h, self.mel_basis.data_ptr(), self.n_mel_channels, y.data_ptr(),
                                    out.data_ptr(), B, N, ws.data_ptr(), ws.numel(), C.c_void_p(stream)))
    return out

  def get_wav_tensor_from_file(self, wav_path) -> torch.Tensor:
    wav, sampling_rate = wav_to_float32(wav_path)
    if sampling_rate != self.sampling_rate:
      raise ValueError(f"{wav_path}: The sampling rate of the file ({sampling_rate}Hz) doesn't match the target "
                       f"sampling rate ({self.sampling_rate}Hz)!")
    return torch.from_numpy(np.ascontiguousarray(wav, dtype=np.float32))

  def get_mel_tensor_from_file(self, wav_path) -> torch.Tensor:
    return self.get_mel_tensor(self.get_wav_tensor_from_file(wav_path))

  def get_mel_tensor(self, wav_tensor: torch.Tensor) -> torch.Tensor:
    return self.mel_spectrogram(wav_tensor.unsqueeze(0)).squeeze(0)
