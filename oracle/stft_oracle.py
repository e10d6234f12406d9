"""CPU restatement of the reference's STFT-based denoiser (TEST INFRASTRUCTURE; numpy fp64).

Follows src/waveglow/stft.py:98-198 (Prem Seetharaman's conv-STFT: Hann-windowed Fourier bases, reflect padding,
magnitude/phase, inverse via the pseudo-inverse basis + window-sum-square normalisation) and
src/waveglow/denoiser.py:51-57 (spectral subtraction).  PARITY UNPINNED: the reference module imports librosa
(absent here) so no golden vectors could be generated from it; ``pad_center`` / ``tiny`` / ``normalize(norm=None)`` are
restated from their documented behaviour, and scipy's ``get_window('hann', n, fftbins=True)`` is the same call the
reference makes (stft.py:125).
"""
import numpy as np
from scipy.signal import get_window


def bases(filter_length=1024, hop_length=256, win_length=1024, window="hann"):
  """forward_basis [2*cutoff, N], inverse_basis [2*cutoff, N] (stft.py:108-132), window^2 [N]."""
  scale = filter_length / hop_length
  fb = np.fft.fft(np.eye(filter_length))
  cutoff = filter_length // 2 + 1
  fb = np.vstack([np.real(fb[:cutoff]), np.imag(fb[:cutoff])])
  inv = np.linalg.pinv(scale * fb).T
  win = get_window(window, win_length, fftbins=True)
  lpad = (filter_length - win_length) // 2
  win = np.pad(win, (lpad, filter_length - win_length - lpad))           # pad_center
  return fb * win, inv * win, win ** 2


def transform(x, fwd, filter_length=1024, hop_length=256):
  """x [B, N] -> (real [B, cutoff, F], imag [B, cutoff, F]) (stft.py:134-163)."""
  xp = np.pad(x, ((0, 0), (filter_length // 2, filter_length // 2)), mode="reflect")
  F = (xp.shape[1] - filter_length) // hop_length + 1
  frames = np.stack([xp[:, f * hop_length:f * hop_length + filter_length] for f in range(F)], axis=2)   # [B, N, F]
  ft = np.einsum("kn,bnf->bkf", fwd, frames)
  cutoff = filter_length // 2 + 1
  return ft[:, :cutoff], ft[:, cutoff:]


def inverse(re, im, inv, win_sq, filter_length=1024, hop_length=256):
  """(stft.py:165-198): conv_transpose with the inverse basis, / window_sumsquare, * N/hop, crop N/2 each side."""
  B, cutoff, F = re.shape
  rec = np.concatenate([re, im], axis=1)                                   # [B, 2*cutoff, F]
  n = filter_length + hop_length * (F - 1)
  out = np.zeros((B, n))
  wsum = np.zeros(n)
  for f in range(F):
    out[:, f * hop_length:f * hop_length + filter_length] += np.einsum("bk,kn->bn", rec[:, :, f], inv)
    wsum[f * hop_length:f * hop_length + filter_length] += win_sq
  nz = wsum > np.finfo(np.float32).tiny
  out[:, nz] /= wsum[nz]
  out *= filter_length / hop_length
  return out[:, filter_length // 2:-(filter_length // 2)]


def denoise(audio, bias_mag, strength, fwd, inv, win_sq):
  """denoiser.py:51-57: magnitude minus strength * bias magnitude (first frame of the bias audio), clamped at 0,
  original phase, inverse STFT."""
  re, im = transform(audio, fwd)
  mag = np.sqrt(re ** 2 + im ** 2)
  ph = np.arctan2(im, re)
  mag_d = np.clip(mag - bias_mag[None, :, None] * strength, 0.0, None)
  return inverse(mag_d * np.cos(ph), mag_d * np.sin(ph), inv, win_sq)


def mel_spectrogram(audio: np.ndarray, mel_basis: np.ndarray, filter_length=1024, hop_length=256, win_length=1024):
  """TacotronSTFT.mel_spectrogram (src/waveglow/taco_stft.py:84-104) in fp64: log(clamp(mel_basis @ |STFT(audio)|, 1e-5)).
  audio [B, N] -> [B, n_mel, N // hop + 1].  Parity unpinned against the reference (librosa absent), see header."""
  fwd, _, _ = bases(filter_length, hop_length, win_length)
  re, im = transform(np.asarray(audio, dtype=np.float64), fwd, filter_length, hop_length)
  mel = np.einsum("mk,bkf->bmf", mel_basis.astype(np.float64), np.sqrt(re ** 2 + im ** 2))
  return np.log(np.maximum(mel, 1e-5))
