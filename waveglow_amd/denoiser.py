"""Bias-removal post-filter (src/waveglow/denoiser.py:14-57, src/waveglow/stft.py:98-198).

NOT part of the HIP hot path: SURVEY.md section 8(f) row 2 ("next").  For now this is host-side torch-op plumbing so
that ``Synthesizer`` is usable; it gets its own HIP kernel and oracle in a later round.  Parity is UNPINNED: the
reference's STFT imports librosa (absent here), so no golden vectors could be generated; ``pad_center`` / ``tiny`` /
``normalize(norm=None)`` are restated from their documented behaviour.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F
from scipy.signal import get_window

BIAS_MEL_LENGTH = 88   # denoiser.py:11


def _pad_center(x: np.ndarray, size: int) -> np.ndarray:
  lpad = (size - len(x)) // 2
  return np.pad(x, (lpad, size - len(x) - lpad))


class STFT(torch.nn.Module):
  def __init__(self, device, filter_length=1024, hop_length=256, win_length=1024, window="hann"):
    super().__init__()
    self.filter_length, self.hop_length, self.win_length, self.window = filter_length, hop_length, win_length, window
    scale = filter_length / hop_length
    fb = np.fft.fft(np.eye(filter_length))
    cutoff = filter_length // 2 + 1
    fb = np.vstack([np.real(fb[:cutoff]), np.imag(fb[:cutoff])])
    fwd = torch.FloatTensor(fb[:, None, :])
    inv = torch.FloatTensor(np.linalg.pinv(scale * fb).T[:, None, :])
    win = torch.from_numpy(_pad_center(get_window(window, win_length, fftbins=True), filter_length)).float()
    self.register_buffer("forward_basis", (fwd * win).float())
    self.register_buffer("inverse_basis", (inv * win).float())
    self.to(device)

  def transform(self, x):
    B, N = x.shape
    x = F.pad(x.view(B, 1, 1, N), (self.filter_length // 2, self.filter_length // 2, 0, 0), mode="reflect").squeeze(1)
    ft = F.conv1d(x, self.forward_basis, stride=self.hop_length)
    cutoff = self.filter_length // 2 + 1
    re, im = ft[:, :cutoff], ft[:, cutoff:]
    return torch.sqrt(re ** 2 + im ** 2), torch.atan2(im, re)

  def inverse(self, magnitude, phase):
    rec = torch.cat([magnitude * torch.cos(phase), magnitude * torch.sin(phase)], dim=1)
    out = F.conv_transpose1d(rec, self.inverse_basis, stride=self.hop_length)
    n_frames = magnitude.size(-1)
    n = self.filter_length + self.hop_length * (n_frames - 1)
    wsq = _pad_center(get_window(self.window, self.win_length, fftbins=True) ** 2, self.filter_length)
    wsum = np.zeros(n, dtype=np.float32)
    for i in range(n_frames):
      s = i * self.hop_length
      wsum[s:min(n, s + self.filter_length)] += wsq[:max(0, min(self.filter_length, n - s))]
    idx = torch.from_numpy(np.where(wsum > np.finfo(np.float32).tiny)[0]).to(out.device)
    wsum_t = torch.from_numpy(wsum).to(out.device)
    out[:, :, idx] /= wsum_t[idx]
    out *= float(self.filter_length) / self.hop_length
    return out[:, :, self.filter_length // 2:-(self.filter_length // 2)]


class Denoiser(torch.nn.Module):
  def __init__(self, waveglow, hparams, mode: str, device):
    super().__init__()
    self.stft = STFT(device, hparams.filter_length, hparams.hop_length, hparams.win_length)
    w = waveglow.upsample.weight
    if mode == "zeros":
      mel = torch.zeros((1, hparams.n_mel_channels, BIAS_MEL_LENGTH), dtype=w.dtype, device=w.device)
    elif mode == "normal":
      mel = torch.randn((1, hparams.n_mel_channels, BIAS_MEL_LENGTH), dtype=w.dtype, device=w.device)
    else:
      raise Exception(f"Mode {mode} if not supported")
    with torch.no_grad():
      bias_audio = waveglow.infer(mel, sigma=0.0).float()          # denoiser.py:45-47 -> the HIP hot path
      bias_spec, _ = self.stft.transform(bias_audio)
    self.register_buffer("bias_spec", bias_spec[:, :, 0][:, :, None])

  def forward(self, audio, strength: float):
    spec, angles = self.stft.transform(audio.float())
    spec = torch.clamp(spec - self.bias_spec * strength, 0.0)
    return self.stft.inverse(spec, angles)
