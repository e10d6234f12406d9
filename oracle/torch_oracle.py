"""CPU fp32 restatement of the reference WaveGlow hot path (TEST INFRASTRUCTURE).

Oracle for ``WaveGlow.infer`` / ``WaveGlow.forward`` / ``WaveGlowLoss`` of
stefantaubert/waveglow, restated functionally over a plain ``dict`` of fp32
tensors keyed by the reference's state_dict names in weight-norm-removed form
(``WN.k.in_layers.i.weight`` ...).  Every function cites the reference lines it
follows (paths relative to /root/reference/).

Pinning: the reference's own tests hold no offline-reachable numeric fixture
for this path (the LJS checkpoint is a network download).
This restatement is pinned instead against outputs of the reference itself,
generated in the build container by ``tests/golden/make_golden.py`` (which
imports the reference's ``model.py``) and committed under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks bit-equality on CPU.

Noise contract: the reference draws three device-RNG
tensors inside ``infer`` -- ``[B,n_rem,L]`` first, then ``[B,n_early,L]`` at
each early-output flow in descending k -- so parity is defined with the noise
injected: ``z_init`` and ``z_early`` (dict: flow index k -> tensor).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


class OracleConfig:
  """Mirror of the model-shaping fields of ModelHParams (src/waveglow/hparams.py:19-31)."""

  def __init__(self, n_mel_channels=80, n_flows=12, n_group=8, n_early_every=4,
               n_early_size=2, n_layers=8, n_channels=256, kernel_size=3,
               upsample_kernel=1024, upsample_stride=256):
    self.n_mel_channels = n_mel_channels
    self.n_flows = n_flows
    self.n_group = n_group
    self.n_early_every = n_early_every
    self.n_early_size = n_early_size
    self.n_layers = n_layers
    self.n_channels = n_channels
    self.kernel_size = kernel_size
    self.upsample_kernel = upsample_kernel      # model.py:148
    self.upsample_stride = upsample_stride      # model.py:149

  def flow_channels(self) -> List[int]:
    """c_k = remaining channels at flow k (model.py:160-176)."""
    out = []
    rem = self.n_group
    for k in range(self.n_flows):
      if k % self.n_early_every == 0 and k > 0:
        rem -= self.n_early_size
      out.append(rem)
    return out

  def early_flows(self) -> List[int]:
    return [k for k in range(self.n_flows) if k % self.n_early_every == 0 and k > 0]


def _as_reference_storage(W3: Tensor) -> Tensor:
  """[c,c,1] weight re-laid column-major, as the reference's parameter is stored: its storage is
  the QR factor assigned at model.py:37-43 (LAPACK order, strides (1,c,1)) and ``load_state_dict``
  copies values into it without changing strides.  LAPACK-backed ``inverse``/``logdet`` round
  differently for the two layouts, so bit-parity needs the same one."""
  return W3.squeeze(-1).t().contiguous().t().unsqueeze(-1)


def squeeze_spect(spect: Tensor, n_group: int) -> Tensor:
  """[B,M,8L] -> [B,M*8,L], channel = mel*8+g (model.py:191-193 / :230-232)."""
  spect = spect.unfold(2, n_group, n_group).permute(0, 2, 1, 3)
  return spect.contiguous().view(spect.size(0), spect.size(1), -1).permute(0, 2, 1)


def upsample_infer(w: Dict[str, Tensor], mel: Tensor, cfg: OracleConfig) -> Tensor:
  """ConvTranspose1d + trim of kernel-stride samples (model.py:225-228)."""
  up = F.conv_transpose1d(mel, w["upsample.weight"], w["upsample.bias"],
                          stride=cfg.upsample_stride)
  cut = cfg.upsample_kernel - cfg.upsample_stride
  return up[:, :, :-cut]


def wn_forward(w: Dict[str, Tensor], k: int, audio_half: Tensor, spect: Tensor,
               cfg: OracleConfig) -> Tensor:
  """WN.forward (model.py:115-138) for flow k; returns [B, 2*h_k, L] = [b ; log_s]."""
  C = cfg.n_channels
  p = f"WN.{k}."
  x = F.conv1d(audio_half, w[p + "start.weight"], w[p + "start.bias"])            # :117
  output = torch.zeros_like(x)                                                     # :118
  cond = F.conv1d(spect, w[p + "cond_layer.weight"], w[p + "cond_layer.bias"])    # :121
  for i in range(cfg.n_layers):
    d = 2 ** i
    pad = int((cfg.kernel_size * d - d) / 2)                                       # :100
    in_res = F.conv1d(x, w[p + f"in_layers.{i}.weight"], w[p + f"in_layers.{i}.bias"],
                      dilation=d, padding=pad)                                     # :125
    a = in_res + cond[:, 2 * C * i:2 * C * (i + 1), :]                             # :16, :126
    acts = torch.tanh(a[:, :C, :]) * torch.sigmoid(a[:, C:, :])                    # :17-19
    rs = F.conv1d(acts, w[p + f"res_skip_layers.{i}.weight"],
                  w[p + f"res_skip_layers.{i}.bias"])                              # :130
    if i < cfg.n_layers - 1:
      x = x + rs[:, :C, :]                                                         # :132
      output = output + rs[:, C:, :]                                               # :133
    else:
      output = output + rs                                                         # :135
  return F.conv1d(output, w[p + "end.weight"], w[p + "end.bias"])                  # :137


def infer_ref(w: Dict[str, Tensor], mel: Tensor, z_init: Tensor,
              z_early: Dict[int, Tensor], sigma: float, cfg: OracleConfig,
              trace: Dict[int, Tensor] | None = None) -> Tensor:
  """WaveGlow.infer (model.py:223-274) with injected noise.  Returns [B, 256*T]."""
  spect = squeeze_spect(upsample_infer(w, mel, cfg), cfg.n_group)
  audio = sigma * z_init                                                           # :243-244
  for k in reversed(range(cfg.n_flows)):                                           # :246
    n_half = audio.size(1) // 2
    a0, a1 = audio[:, :n_half, :], audio[:, n_half:, :]
    out = wn_forward(w, k, a0, spect, cfg)                                         # :251
    s, b = out[:, n_half:, :], out[:, :n_half, :]                                  # :253-254
    a1 = (a1 - b) / torch.exp(s)                                                   # :255
    audio = torch.cat([a0, a1], 1)
    W = _as_reference_storage(w[f"convinv.{k}.conv.weight"]).squeeze()           # :49
    W_inv = W.float().inverse()                                                    # :54
    audio = F.conv1d(audio, W_inv[..., None])                                      # :59
    if k % cfg.n_early_every == 0 and k > 0:                                       # :260
      audio = torch.cat((sigma * z_early[k], audio), 1)                            # :271
    if trace is not None:
      trace[k] = audio.clone()
  return audio.permute(0, 2, 1).contiguous().view(audio.size(0), -1)               # :273


def forward_ref(w: Dict[str, Tensor], mel: Tensor, audio: Tensor, cfg: OracleConfig
                ) -> Tuple[Tensor, List[Tensor], List[Tensor]]:
  """WaveGlow.forward (model.py:178-221): returns (z [B,8,L], [log_s]*F, [log_det_W]*F)."""
  spect = F.conv_transpose1d(mel, w["upsample.weight"], w["upsample.bias"],
                             stride=cfg.upsample_stride)                           # :186
  assert spect.size(2) >= audio.size(1)                                            # :187
  if spect.size(2) > audio.size(1):
    spect = spect[:, :, :audio.size(1)]                                            # :189
  spect = squeeze_spect(spect, cfg.n_group)
  audio = audio.unfold(1, cfg.n_group, cfg.n_group).permute(0, 2, 1)               # :195
  outs, log_s_list, log_det_list = [], [], []
  for k in range(cfg.n_flows):
    if k % cfg.n_early_every == 0 and k > 0:                                       # :201-203
      outs.append(audio[:, :cfg.n_early_size, :])
      audio = audio[:, cfg.n_early_size:, :]
    Wk = _as_reference_storage(w[f"convinv.{k}.conv.weight"])
    B_, _, L_ = audio.size()
    log_det_W = B_ * L_ * torch.logdet(Wk.squeeze())                               # :49, :63
    audio = F.conv1d(audio, Wk)                                                    # :64
    log_det_list.append(log_det_W)
    n_half = audio.size(1) // 2
    a0, a1 = audio[:, :n_half, :], audio[:, n_half:, :]
    out = wn_forward(w, k, a0, spect, cfg)
    log_s, b = out[:, n_half:, :], out[:, :n_half, :]                              # :213-214
    a1 = torch.exp(log_s) * a1 + b                                                 # :215
    log_s_list.append(log_s)
    audio = torch.cat([a0, a1], 1)
  outs.append(audio)
  return torch.cat(outs, 1), log_s_list, log_det_list


def loss_ref(z: Tensor, log_s_list: Sequence[Tensor], log_det_list: Sequence[Tensor],
             sigma: float = 1.0) -> Tensor:
  """WaveGlowLoss.forward (src/waveglow/train.py:31-45)."""
  log_s_total = sum(torch.sum(ls) for ls in log_s_list)
  log_det_total = sum(log_det_list)
  loss = torch.sum(z * z) / (2 * sigma * sigma) - log_s_total - log_det_total
  return loss / (z.size(0) * z.size(1) * z.size(2))


def grads_ref(sd_weightnorm: Dict[str, Tensor], mel: Tensor, audio: Tensor, cfg: OracleConfig, sigma: float = 1.0
              ) -> Tuple[Tensor, Dict[str, Tensor]]:
  """One training step's loss and parameter gradients, as the reference computes them (train.py:190-196:
  forward -> WaveGlowLoss -> backward) on the weight-normed parameter set (686-key form): the weights are
  re-composed with torch._weight_norm(v, g, 0) exactly like the parametrization does, then forward_ref / loss_ref
  run under autograd.  Returns (loss, {state_dict key: gradient})."""
  leaves = {k: v.clone().requires_grad_(True) for k, v in sd_weightnorm.items()}
  dense: Dict[str, Tensor] = {}
  for k, v in leaves.items():
    if k.endswith("parametrizations.weight.original1"):
      base = k[:-len("parametrizations.weight.original1")]
      dense[base + "weight"] = torch._weight_norm(v, leaves[base + "parametrizations.weight.original0"], 0)
    elif not k.endswith("parametrizations.weight.original0"):
      dense[k] = v
  z, log_s, log_det = forward_ref(dense, mel, audio, cfg)
  loss = loss_ref(z, log_s, log_det, sigma)
  grads = torch.autograd.grad(loss, list(leaves.values()))
  return loss.detach(), dict(zip(leaves.keys(), grads))
