"""Bias-removal post-filter with the reference's surface (src/waveglow/denoiser.py:14-57), computed by the HIP
library (``wg_stft_*``: conv-STFT of src/waveglow/stft.py:98-198 as exact-fp32 MFMA GEMMs).

The Fourier / pseudo-inverse bases are built on the host exactly like ``STFT.__init__`` (stft.py:108-132) and handed
to the library once.  librosa's ``pad_center`` is restated (the reference imports it; librosa is absent here), so the
parity of this post-filter is pinned only against ``oracle/stft_oracle.py`` (numpy fp64 restatement), not against
reference outputs.  No torch/CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
from scipy.signal import get_window

from . import _lib

BIAS_MEL_LENGTH = 88   # denoiser.py:11


def stft_bases(filter_length=1024, hop_length=256, win_length=1024, window="hann"):
  """forward basis, inverse basis [2*(N/2+1), N] and squared window [N] (stft.py:108-132, :45-95)."""
  scale = filter_length / hop_length
  fb = np.fft.fft(np.eye(filter_length))
  cutoff = filter_length // 2 + 1
  fb = np.vstack([np.real(fb[:cutoff]), np.imag(fb[:cutoff])])
  inv = np.linalg.pinv(scale * fb).T
  win = get_window(window, win_length, fftbins=True)
  lpad = (filter_length - win_length) // 2
  win = np.pad(win, (lpad, filter_length - win_length - lpad))          # librosa.util.pad_center
  # C-contiguous copies: the pinv transpose is a Fortran-ordered view and the library reads raw row-major memory
  return (np.ascontiguousarray(fb * win, dtype=np.float32), np.ascontiguousarray(inv * win, dtype=np.float32),
          np.ascontiguousarray(win ** 2, dtype=np.float32))


class Denoiser(torch.nn.Module):
  """Removes model bias from audio produced with waveglow (denoiser.py:14-57)."""

  def __init__(self, waveglow, hparams, mode: str, device):
    super().__init__()
    device = torch.device(device)
    if device.type != "cuda":
      raise _lib.WgError("the denoiser runs on the GPU library only")
    self.lib = _lib.load()
    fwd, inv, wsq = stft_bases(hparams.filter_length, hparams.hop_length, hparams.win_length)
    self._h = C.c_void_p()
    _lib.check(self.lib.wg_stft_create(fwd.ctypes.data, inv.ctypes.data, wsq.ctypes.data, hparams.filter_length,
                                       hparams.hop_length, _lib.device_index(device), C.byref(self._h)))
    w = waveglow.upsample.weight
    if mode == "zeros":
      mel = torch.zeros((1, hparams.n_mel_channels, BIAS_MEL_LENGTH), dtype=w.dtype, device=w.device)
    elif mode == "normal":
      mel = torch.randn((1, hparams.n_mel_channels, BIAS_MEL_LENGTH), dtype=w.dtype, device=w.device)
    else:
      raise Exception(f"Mode {mode} if not supported")
    with torch.no_grad():
      bias_audio = waveglow.infer(mel, sigma=0.0).float()              # denoiser.py:45-47 (the HIP hot path)
      mag0 = torch.empty((1, hparams.filter_length // 2 + 1), dtype=torch.float32, device=bias_audio.device)
      self._run(bias_audio, None, 0.0, None, mag0)
    self.register_buffer("bias_spec", mag0[:, :, None])                # [1, 513, 1] like bias_spec[:, :, 0][:, :, None]

  def __del__(self):
    try:
      if getattr(self, "_h", None):
        self.lib.wg_stft_destroy(self._h)
    except Exception:
      pass

  def _run(self, audio, bias, strength, out, mag0):
    audio = audio.contiguous()
    B, N = audio.shape
    nbytes = self.lib.wg_stft_workspace_bytes(self._h, B, N)
    if nbytes == 0:
      raise _lib.WgError(f"denoiser: unsupported audio length {N} (multiple of 256, >= 1024)")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=audio.device)
    stream = torch.cuda.current_stream(audio.device).cuda_stream
    _lib.check(self.lib.wg_stft_denoise(self._h, audio.data_ptr(), bias.data_ptr() if bias is not None else None,
                                        float(strength), out.data_ptr() if out is not None else None,
                                        mag0.data_ptr() if mag0 is not None else None, B, N, ws.data_ptr(),
                                        ws.numel(), C.c_void_p(stream)))

  def forward(self, audio: torch.Tensor, strength: float):
    """denoiser.py:51-57; returns [B, 1, N] like the reference's conv_transpose1d output."""
    audio = audio.float()
    out = torch.empty_like(audio)
    self._run(audio, self.bias_spec.reshape(-1).contiguous(), strength, out, None)
    return out[:, None, :]
