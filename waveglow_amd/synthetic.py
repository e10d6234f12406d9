"""Deterministic synthetic checkpoints and inputs (no network: the LJS checkpoint of
the reference, src/waveglow/dl_pretrained.py:28-43, is a download).

Weights are generated per state_dict key from ``crc32(key) ^ seed`` so any subset
can be regenerated independently on any machine; shapes and key names are the
reference's (src/waveglow/model.py:141-176, weight-norm-removed form).  ``end`` is
non-zero on purpose: the reference zero-initialises it (model.py:90-92), which
turns every coupling into the identity and would hide bugs.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, List, Tuple

import torch

from .hparams import HParams


def _gen(key: str, seed: int) -> torch.Generator:
  g = torch.Generator(device="cpu")
  g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
  return g


def _normal(key: str, seed: int, shape, std: float) -> torch.Tensor:
  return torch.empty(*shape, dtype=torch.float32).normal_(0.0, std, generator=_gen(key, seed))


def flow_channels(hp: HParams) -> List[int]:
  out, rem = [], hp.n_group
  for k in range(hp.n_flows):
    if k % hp.n_early_every == 0 and k > 0:
      rem -= hp.n_early_size
    out.append(rem)
  return out


def make_state_dict(hp: HParams, seed: int = 0) -> Dict[str, torch.Tensor]:
  """470-key-style (weight-norm removed) fp32 state dict with reference key names."""
  C, M, G = hp.n_channels, hp.n_mel_channels, hp.n_group
  sd: Dict[str, torch.Tensor] = {}
  up_k = 1024
  sd["upsample.weight"] = _normal("upsample.weight", seed, (M, M, up_k), 1.0 / (5.4 * math.sqrt(4 * M)))
  sd["upsample.bias"] = _normal("upsample.bias", seed, (M,), 0.05)
  for k, ck in enumerate(flow_channels(hp)):
    h = ck // 2
    key = f"convinv.{k}.conv.weight"
    q = torch.linalg.qr(_normal(key, seed, (ck, ck), 1.0))[0]
    if torch.det(q) < 0:
      q[:, 0] = -q[:, 0]
    sd[key] = q.contiguous().view(ck, ck, 1)
    p = f"WN.{k}."
    sd[p + "start.weight"] = _normal(p + "start.weight", seed, (C, h, 1), 1.0 / math.sqrt(h))
    sd[p + "start.bias"] = _normal(p + "start.bias", seed, (C,), 0.1)
    sd[p + "cond_layer.weight"] = _normal(p + "cond_layer.weight", seed,
                                          (2 * C * hp.n_layers, M * G, 1), 0.7 / math.sqrt(M * G))
    sd[p + "cond_layer.bias"] = _normal(p + "cond_layer.bias", seed, (2 * C * hp.n_layers,), 0.1)
    for i in range(hp.n_layers):
      kk = p + f"in_layers.{i}."
      sd[kk + "weight"] = _normal(kk + "weight", seed, (2 * C, C, hp.kernel_size),
                                  0.7 / math.sqrt(hp.kernel_size * C))
      sd[kk + "bias"] = _normal(kk + "bias", seed, (2 * C,), 0.1)
      rs = 2 * C if i < hp.n_layers - 1 else C
      kk = p + f"res_skip_layers.{i}."
      sd[kk + "weight"] = _normal(kk + "weight", seed, (rs, C, 1), 1.0 / math.sqrt(C))
      sd[kk + "bias"] = _normal(kk + "bias", seed, (rs,), 0.05)
    sd[p + "end.weight"] = _normal(p + "end.weight", seed, (2 * h, C, 1), 0.2 / math.sqrt(C))
    sd[p + "end.bias"] = _normal(p + "end.bias", seed, (2 * h,), 0.02)
  return sd


def to_weightnorm_form(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
  """Re-express dense weights as the 686-key checkpoint form (``parametrizations.weight.original0/1``,
  g = per-output-channel norm, v = w) that the reference saves while training."""
  out: Dict[str, torch.Tensor] = {}
  for key, val in sd.items():
    parts = key.split(".")
    normed = (parts[0] == "WN" and parts[-1] == "weight" and parts[2] != "end")
    if normed:
      base = key[:-len("weight")]
      g = val.flatten(1).norm(dim=1).view(-1, 1, 1)
      out[base + "parametrizations.weight.original0"] = g
      out[base + "parametrizations.weight.original1"] = val.clone()
    else:
      out[key] = val
  return out


def make_mel(B: int, T: int, n_mel: int = 80, seed: int = 1234) -> torch.Tensor:
  """Synthetic log-mel: N(-5,2) clamped to [-11.5, 2] (floor = log(1e-5), taco_stft.py:10-16)."""
  g = torch.Generator(device="cpu")
  g.manual_seed(seed)
  return torch.empty(B, n_mel, T).normal_(-5.0, 2.0, generator=g).clamp_(-11.5, 2.0)


def make_noise(hp: HParams, B: int, L: int, seed: int = 4321) -> Tuple[torch.Tensor, Dict[int, torch.Tensor]]:
  """z_init [B,n_rem,L] and z_early {k: [B,n_early,L]} in the reference's draw order
  (model.py:234-244 then :260-271 for descending k)."""
  g = torch.Generator(device="cpu")
  g.manual_seed(seed)
  n_rem = flow_channels(hp)[-1]
  z_init = torch.empty(B, n_rem, L).normal_(generator=g)
  z_early = {}
  for k in reversed(range(hp.n_flows)):
    if k % hp.n_early_every == 0 and k > 0:
      z_early[k] = torch.empty(B, hp.n_early_size, L).normal_(generator=g)
  return z_init, z_early
