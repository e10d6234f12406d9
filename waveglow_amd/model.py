"""``WaveGlow`` with the reference's Python surface, computing on the MI355X-native HIP library.

Mirrors src/waveglow/model.py of stefantaubert/waveglow: same constructor argument, same parameter /
state_dict names and shapes (so ``CheckpointWaveglow.load`` -> ``load_state_dict`` works unchanged, incl.
the weight-normed 686-key form, the 470-key form after ``remove_weightnorm`` and legacy
``weight_g``/``weight_v`` keys), same ``infer`` / ``forward`` / ``remove_weightnorm`` signatures.

The torch modules below only HOLD parameters.  All arithmetic of ``infer`` and ``forward`` runs in
``libwaveglow_amd.so`` (waveglow_amd/csrc) through its C ABI; there is no torch/CPU fallback -- a CPU
tensor or a missing library raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from . import _lib
from .hparams import HParams


class Invertible1x1Conv(nn.Module):
  """Parameter holder for the invertible 1x1 convolution (model.py:23-43): QR-orthonormal init, det +1."""

  def __init__(self, c: int):
    super().__init__()
    self.conv = nn.Conv1d(c, c, kernel_size=1, stride=1, padding=0, bias=False)
    W = torch.linalg.qr(torch.empty(c, c).normal_())[0]
    if torch.det(W) < 0:
      W[:, 0] = -1 * W[:, 0]
    self.conv.weight.data = W.contiguous().view(c, c, 1)


class WN(nn.Module):
  """Parameter holder for the WaveNet-like coupling network (model.py:75-113)."""

  def __init__(self, n_in_channels: int, n_mel_channels: int, hparams: HParams):
    super().__init__()
    assert hparams.kernel_size % 2 == 1
    assert hparams.n_channels % 2 == 0
    self.n_layers = hparams.n_layers
    self.n_channels = hparams.n_channels
    wn = nn.utils.parametrizations.weight_norm
    self.in_layers = nn.ModuleList()
    self.res_skip_layers = nn.ModuleList()
    self.start = wn(nn.Conv1d(n_in_channels, self.n_channels, 1), name="weight")
    end = nn.Conv1d(self.n_channels, 2 * n_in_channels, 1)
    end.weight.data.zero_()   # model.py:90-92
    end.bias.data.zero_()
    self.end = end
    self.cond_layer = wn(nn.Conv1d(n_mel_channels, 2 * self.n_channels * self.n_layers, 1), name="weight")
    for i in range(self.n_layers):
      dilation = 2 ** i
      padding = int((hparams.kernel_size * dilation - dilation) / 2)
      self.in_layers.append(wn(nn.Conv1d(self.n_channels, 2 * self.n_channels, hparams.kernel_size,
                                         dilation=dilation, padding=padding), name="weight"))
      rs = 2 * self.n_channels if i < self.n_layers - 1 else self.n_channels
      self.res_skip_layers.append(wn(nn.Conv1d(self.n_channels, rs, 1), name="weight"))


def _dense_weight(conv: nn.Module) -> torch.Tensor:
  """``conv.weight`` -- for a weight-normed module this evaluates g*v/||v|| exactly as torch does."""
  return conv.weight


class _Engine:
  """Owns one wg_handle and the packed device weights derived from a module's parameters."""

  KERNEL_WIDTHS = (64, 128, 256, 512)          # n_channels the WN-layer kernel is instantiated for

  def __init__(self, hp: HParams, device: torch.device):
    lib = _lib.load()
    # Any n_channels up to 512 (the reference takes any even number, model.py:78): the kernels run at the next
    # instantiated width with the extra channels' weights and biases zero -- x_pad = 0, tanh(0) * sigmoid(0) = 0, res / skip
    # add nothing: the results are exactly those of the unpadded network (inference and the no-grad forward; training
    # takes the instantiated widths only).
    self.n_channels = hp.n_channels
    self.width = next((w for w in self.KERNEL_WIDTHS if w >= hp.n_channels), 0)
    if not self.width or hp.n_channels < 1:
      raise _lib.WgError(f"n_channels={hp.n_channels} unsupported (1..512)")
    self.n_layers = hp.n_layers
    # the same for the mel axis: the conditioning K-steps take multiples of 16 mel channels (<= 80); other counts run with
    # zero filter taps / conditioning columns for the extra channels and a zero-padded input
    self.n_mel = hp.n_mel_channels
    self.mel_width = (hp.n_mel_channels + 15) // 16 * 16
    if not 1 <= hp.n_mel_channels <= 80:
      raise _lib.WgError(f"n_mel_channels={hp.n_mel_channels} unsupported (1..80)")
    cfg = _lib.WgConfig(self.mel_width, hp.n_flows, hp.n_group, hp.n_early_every, hp.n_early_size,
                        hp.n_layers, self.width, hp.kernel_size, 1024, 256)
    handle = C.c_void_p()
    _lib.check(lib.wg_create(C.byref(cfg), _lib.device_index(device), C.byref(handle)))
    self.lib = lib
    self.handle = handle
    self.device = device
    self.signature: Optional[tuple] = None
    self._ws: Dict[Tuple[str, int, int, int], torch.Tensor] = {}
    self._train_pool: List[dict] = []          # training workspaces (saved activations), see train_workspace
    self._graphs: Dict[tuple, tuple] = {}      # captured hipGraphs of wg_infer, keyed by input shape / dtype / sigma
    self._graphs_sig: Optional[tuple] = None   # weight signature the graphs were captured with
    self.cache: Dict[tuple, object] = {}       # small per-handle lookups (canonical parameter names of the training direction)

  def __del__(self):
    try:
      if getattr(self, "handle", None):
        self.lib.wg_destroy(self.handle)
    except Exception:
      pass

  def upload(self, tensors: Dict[str, torch.Tensor]) -> None:
    n = self.lib.wg_num_expected_tensors(self.handle)
    for i in range(n):
      name = self.lib.wg_expected_tensor_name(self.handle, i).decode()
      if name not in tensors:
        raise _lib.WgError(f"missing weight '{name}'")
      t = tensors[name].detach().to(device="cpu", dtype=torch.float32)
      if self.width != self.n_channels:
        t = self._pad_channels(name, t)
      if self.mel_width != self.n_mel:
        t = self._pad_mel(name, t)
      t = t.contiguous()
      shape = (C.c_int64 * t.dim())(*t.shape)
      _lib.check(self.lib.wg_set_tensor(self.handle, name.encode(), C.c_void_p(t.data_ptr()), shape, t.dim()))
    _lib.check(self.lib.wg_finalize(self.handle))

  def _pad_channels(self, name: str, t: torch.Tensor) -> torch.Tensor:
    """Zero-pad the WN channel axes of one dense (470-key form) tensor from n_channels to the kernel width.  The gate
    halves (tanh rows | sigmoid rows, model.py:13-20) and the res | skip halves (model.py:131-136) are padded separately."""
    Cc, W, nl = self.n_channels, self.width, self.n_layers
    pad = torch.nn.functional.pad
    parts = name.split(".")
    if parts[0] != "WN":
      return t
    mod, leaf = parts[2], parts[-1]

    def halves(x, n_halves):            # [n_halves * Cc, ...] -> [n_halves * W, ...]
      x = x.reshape(n_halves, Cc, *x.shape[1:])
      return pad(x, (0, 0) * (x.dim() - 2) + (0, W - Cc)).reshape(n_halves * W, *x.shape[2:])

    if mod == "start":
      return halves(t, 1)
    if mod == "end":
      return pad(t, (0, 0, 0, W - Cc)) if leaf == "weight" else t            # [2h, C, 1]: the input-channel axis
    if mod == "cond_layer":
      return halves(t, 2 * nl)
    if mod == "in_layers":
      t = halves(t, 2)
      return pad(t, (0, 0, 0, W - Cc)) if leaf == "weight" else t            # [2W, C, 3] -> [2W, W, 3]
    if mod == "res_skip_layers":
      t = halves(t, t.shape[0] // Cc)
      return pad(t, (0, 0, 0, W - Cc)) if leaf == "weight" else t
    return t

  def _pad_mel(self, name: str, t: torch.Tensor) -> torch.Tensor:
    """Zero-pad the mel axes from n_mel_channels to the next multiple of 16: upsample [M, M, 1024] / [M], and the input
    axis of cond_layer [.., 8 M, 1], whose columns are mel-major (column = mel * 8 + g: the padding is a tail)."""
    pad = torch.nn.functional.pad
    d = self.mel_width - self.n_mel
    if name == "upsample.weight":
      return pad(t, (0, 0, 0, d, 0, d))
    if name == "upsample.bias":
      return pad(t, (0, d))
    if name.endswith("cond_layer.weight"):
      return pad(t, (0, 0, 0, 8 * d))
    return t

  def pad_mel_input(self, spect: torch.Tensor) -> torch.Tensor:
    if self.mel_width == self.n_mel:
      return spect
    return torch.nn.functional.pad(spect, (0, 0, 0, self.mel_width - self.n_mel))

  def workspace(self, kind: str, nbytes: int, key: Tuple[int, int, int]) -> torch.Tensor:
    """One cached workspace (the last shape's).  A captured hipGraph bakes its workspace's device pointer in, so
    ``_infer_graphed`` keeps its own reference in the graph's cache entry: evicting here never frees memory that a
    graph will still write to."""
    k = (kind,) + key
    ws = self._ws.get(k)
    if ws is None or ws.numel() < nbytes:
      self._ws.clear()
      ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
      self._ws[k] = ws
    return ws


  def train_workspace(self, nbytes: int, key: Tuple[int, ...]) -> Tuple[dict, bool]:
    """A workspace of the training direction (saved activations) that no pending backward still needs.
    ``fresh`` tells the library to clear it (guard rows must read as zero; they stay zero as long as the geometry
    ``key`` does not change).  Normally there is exactly one; a second forward() before the first backward()
    (gradient accumulation over micro-batches, two losses) gets another one instead of clobbering the first."""
    pool = self._train_pool
    pool[:] = [e for e in pool if e["key"] == key and e["ws"].numel() >= nbytes]   # geometry changed: drop the old ones
    for e in pool:
      if not e["busy"]:
        e["busy"] = True
        return e, False
    e = {"key": key, "ws": torch.empty(nbytes, dtype=torch.uint8, device=self.device), "busy": True}
    pool.append(e)
    return e, True


class WaveGlow(nn.Module):
  def __init__(self, hparams: HParams):
    super().__init__()
    self.upsample = nn.ConvTranspose1d(hparams.n_mel_channels, hparams.n_mel_channels, 1024, stride=256)
    assert hparams.n_group % 2 == 0
    self.n_flows = hparams.n_flows
    self.n_group = hparams.n_group
    self.n_early_every = hparams.n_early_every
    self.n_early_size = hparams.n_early_size
    self.WN = nn.ModuleList()
    self.convinv = nn.ModuleList()
    n_half = int(self.n_group / 2)
    n_remaining_channels = self.n_group
    for k in range(self.n_flows):
      if k % self.n_early_every == 0 and k > 0:
        n_half = n_half - int(self.n_early_size / 2)
        n_remaining_channels = n_remaining_channels - self.n_early_size
      self.convinv.append(Invertible1x1Conv(n_remaining_channels))
      self.WN.append(WN(n_in_channels=n_half, n_mel_channels=hparams.n_mel_channels * self.n_group,
                        hparams=hparams))
    self.n_remaining_channels = n_remaining_channels
    self._hp = hparams
    self._engine: Optional[_Engine] = None

  # ------------------------------------------------------------------ engine plumbing
  def dense_state(self) -> Dict[str, torch.Tensor]:
    """Weight-norm-folded tensors under the 470-key names (what remove_weightnorm would leave)."""
    out = {"upsample.weight": self.upsample.weight, "upsample.bias": self.upsample.bias}
    for k in range(self.n_flows):
      out[f"convinv.{k}.conv.weight"] = self.convinv[k].conv.weight
      wn, p = self.WN[k], f"WN.{k}."
      for name, mod in (("start", wn.start), ("cond_layer", wn.cond_layer), ("end", wn.end)):
        out[p + name + ".weight"] = _dense_weight(mod)
        out[p + name + ".bias"] = mod.bias
      for i in range(wn.n_layers):
        out[p + f"in_layers.{i}.weight"] = _dense_weight(wn.in_layers[i])
        out[p + f"in_layers.{i}.bias"] = wn.in_layers[i].bias
        out[p + f"res_skip_layers.{i}.weight"] = _dense_weight(wn.res_skip_layers[i])
        out[p + f"res_skip_layers.{i}.bias"] = wn.res_skip_layers[i].bias
    return out

  def _weights_signature(self) -> tuple:
    return tuple((id(p), p._version, p.data_ptr()) for p in self.parameters())

  def flow_channels(self) -> List[int]:
    """Remaining channels c_k of every flow (model.py:160-176)."""
    out, c = [], self.n_group
    for k in range(self.n_flows):
      if k % self.n_early_every == 0 and k > 0:
        c -= self.n_early_size
      out.append(c)
    return out

  def _get_engine(self, device: torch.device, need_weights: bool = True) -> _Engine:
    if device.type != "cuda":
      raise _lib.WgError("waveglow_amd runs on MI355X only: move the model and inputs to a 'cuda' (ROCm) device; "
                         "there is no CPU fallback")
    if device.index is None:
      device = torch.device("cuda", torch.cuda.current_device())
    if self._engine is None or self._engine.device != device:
      self._engine = _Engine(self._hp, device)
    if not need_weights:     # training direction: weights are handed over per call (waveglow_amd/train.py)
      return self._engine
    sig = self._weights_signature()
    if self._engine.signature != sig:
      # W^-1 and every packed layout are derived state keyed on the parameter versions
      # (the reference caches W_inverse as a plain attribute and goes stale: model.py:52-58)
      with torch.no_grad():
        self._engine.upload(self.dense_state())
      self._engine.signature = sig
    return self._engine

  @staticmethod
  def _io_dtype(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
      return _lib.WG_F32
    if t.dtype == torch.float16:
      return _lib.WG_F16
    raise _lib.WgError(f"unsupported dtype {t.dtype} (float32 or float16)")

  # ------------------------------------------------------------------ reference API
  def infer_with_noise(self, spect: torch.Tensor, z_init: torch.Tensor, z_early: List[torch.Tensor],
                       sigma: float = 1.0, frames: Optional[torch.Tensor] = None, graph: bool = False) -> torch.Tensor:
    """``infer`` with the three noise draws injected (z_early in descending flow order).
    ``frames`` (int32 [B], optional): mel-frame count of every utterance of a padded batch -- each utterance then gets
    exactly its batch-of-one result (``wg_infer_ragged``); the audio behind ``256 * frames[b]`` is zero.
    ``graph=True`` replays the call's ~110 kernel launches from a captured hipGraph (one graph per input shape, dtype
    and sigma; the inputs are copied into the graph's static buffers) -- for single utterances the launch gaps are a
    measurable part of the latency.  Results are bit-identical to the direct launches."""
    if graph:
      return self._infer_graphed(spect, z_init, z_early, sigma, frames)
    eng = self._get_engine(spect.device)
    io = self._io_dtype(spect)
    spect = eng.pad_mel_input(spect).contiguous()
    B, M, T = spect.shape
    L = T * 256 // self.n_group
    assert z_init.shape == (B, self.n_remaining_channels, L) and z_init.dtype == spect.dtype
    z_init = z_init.contiguous()
    z_early = [z.contiguous() for z in z_early]
    for z in z_early:
      assert z.shape == (B, self.n_early_size, L) and z.dtype == spect.dtype and z.device == spect.device
    audio = torch.empty((B, T * 256), dtype=spect.dtype, device=spect.device)
    nbytes = eng.lib.wg_infer_workspace_bytes(eng.handle, B, T)
    ws = eng.workspace("infer", nbytes, (B, T, 0))
    ze = (C.c_void_p * max(1, len(z_early)))(*[z.data_ptr() for z in z_early])
    stream = torch.cuda.current_stream(spect.device).cuda_stream
    fr = None
    if frames is not None:
      fr = frames.to(device=spect.device, dtype=torch.int32).contiguous()
      assert fr.shape == (B,)
      if not torch.cuda.is_current_stream_capturing():
        assert int(fr.min()) >= 1 and int(fr.max()) <= T
    _lib.check(eng.lib.wg_infer_ragged(eng.handle, spect.data_ptr(), fr.data_ptr() if fr is not None else None,
                                       z_init.data_ptr(), ze, len(z_early), float(sigma), audio.data_ptr(), B, T, io,
                                       ws.data_ptr(), ws.numel(), C.c_void_p(stream)))
    return audio

  def _infer_graphed(self, spect, z_init, z_early, sigma, frames):
    eng = self._get_engine(spect.device)        # re-uploads weights if they changed: then the old graphs are stale
    cache = eng._graphs
    if eng._graphs_sig != eng.signature:
      cache.clear()
      eng._graphs_sig = eng.signature
    key = (tuple(spect.shape), spect.dtype, float(sigma), len(z_early), frames is not None)
    ent = cache.get(key)
    ins = [spect, z_init] + list(z_early) + ([frames.to(spect.device, torch.int32)] if frames is not None else [])
    if ent is None:
      st = [torch.empty_like(t) for t in ins]
      for dst, src in zip(st, ins):
        dst.copy_(src)
      nz = len(z_early)
      call = lambda: self.infer_with_noise(st[0], st[1], st[2:2 + nz], sigma, frames=st[2 + nz] if frames is not None else None)
      side = torch.cuda.Stream(device=spect.device)
      side.wait_stream(torch.cuda.current_stream(spect.device))
      with torch.cuda.stream(side):             # warm-up outside capture: workspace allocation, lazy kernel loading
        call()
      torch.cuda.current_stream(spect.device).wait_stream(side)
      B_, _, T_ = spect.shape
      ws = eng._ws.get(("infer", B_, T_, 0))    # allocated by the warm-up; the graph's kernels write to THIS memory
      assert ws is not None
      g = torch.cuda.CUDAGraph()
      with torch.cuda.graph(g):
        out = call()
      assert eng._ws.get(("infer", B_, T_, 0)) is ws
      ent = (g, st, out, ws)                    # ws: kept alive as long as the graph (see _Engine.workspace)
      cache[key] = ent
    g, st, out, _ws = ent
    for dst, src in zip(st, ins):
      dst.copy_(src)
    g.replay()
    return out.clone()

  def infer(self, spect: torch.Tensor, sigma: float = 1.0) -> torch.Tensor:
    """model.py:223-274.  Noise is drawn with the device RNG in the tensor dtype, in the reference's
    order: [B,n_rem,L], then [B,n_early,L] per early-output flow for descending k."""
    B, _, T = spect.shape
    L = T * 256 // self.n_group
    z_init = torch.empty((B, self.n_remaining_channels, L), dtype=spect.dtype, device=spect.device).normal_()
    z_early = []
    for k in reversed(range(self.n_flows)):
      if k % self.n_early_every == 0 and k > 0:
        z_early.append(torch.empty((B, self.n_early_size, L), dtype=spect.dtype, device=spect.device).normal_())
    return self.infer_with_noise(spect, z_init, z_early, sigma)

  def forward(self, forward_input):
    """model.py:178-221: (mel [B,M,F], audio [B,S]) -> (z [B,8,L], [log_s_k], [log_det_W_k]).
    With grad mode on and trainable parameters this is the training direction (waveglow_amd/train.py: saved
    activations, ``loss.backward()`` runs the library's backward pass); otherwise the lighter inference-only pass."""
    spect, audio = forward_input
    if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
      from .train import train_forward
      # grad_scale: loss scale of the fp16 gradient planes (0 = automatic, 2^round(log2 N) for the reference's mean loss)
      return train_forward(self, spect, audio, float(getattr(self, "grad_scale", 0.0)))
    eng = self._get_engine(spect.device)
    io = self._io_dtype(spect)
    assert audio.dtype == spect.dtype and audio.device == spect.device
    spect, audio = eng.pad_mel_input(spect).contiguous(), audio.contiguous()
    B, M, F_ = spect.shape
    S = audio.shape[1]
    assert (F_ - 1) * 256 + 1024 >= S   # model.py:187
    S = S - S % self.n_group            # unfold drops the remainder (model.py:191,195)
    if S != audio.shape[1]:
      audio = audio[:, :S].contiguous()
    L = S // self.n_group
    z = torch.empty((B, self.n_group, L), dtype=torch.float32, device=spect.device)
    log_s, c = [], self.n_group
    for k in range(self.n_flows):
      if k % self.n_early_every == 0 and k > 0:
        c -= self.n_early_size
      log_s.append(torch.empty((B, c // 2, L), dtype=torch.float32, device=spect.device))
    log_det = (C.c_float * self.n_flows)()
    nbytes = eng.lib.wg_forward_workspace_bytes(eng.handle, B, F_, S)
    ws = eng.workspace("fwd", nbytes, (B, F_, S))
    ls = (C.c_void_p * self.n_flows)(*[t.data_ptr() for t in log_s])
    stream = torch.cuda.current_stream(spect.device).cuda_stream
    _lib.check(eng.lib.wg_forward(eng.handle, spect.data_ptr(), audio.data_ptr(), z.data_ptr(), ls, log_det,
                                  B, F_, S, io, ws.data_ptr(), ws.numel(), C.c_void_p(stream)))
    log_det_list = [torch.tensor(log_det[k], dtype=torch.float32, device=spect.device) for k in range(self.n_flows)]
    if spect.dtype != torch.float32:
      z = z.to(spect.dtype)
      log_s = [t.to(spect.dtype) for t in log_s]
    return z, log_s, log_det_list

  @staticmethod
  def remove_weightnorm(model: "WaveGlow") -> "WaveGlow":
    """model.py:276-297: materialise w = g*v/||v|| for start, cond_layer, in_layers, res_skip_layers."""
    rp = nn.utils.parametrize.remove_parametrizations
    for wnet in model.WN:
      if nn.utils.parametrize.is_parametrized(wnet.start, "weight"):
        wnet.start = rp(wnet.start, "weight")
        wnet.cond_layer = rp(wnet.cond_layer, "weight")
        wnet.in_layers = nn.ModuleList([rp(m, "weight") for m in wnet.in_layers])
        wnet.res_skip_layers = nn.ModuleList([rp(m, "weight") for m in wnet.res_skip_layers])
    return model


class _LossFn(torch.autograd.Function):
  """Value from ``wg_loss``; gradient analytically: d/dz = z/(sigma^2 N), d/dlog_s = -1/N, d/dlog_det_W = -1/N."""

  @staticmethod
  def forward(ctx, sigma, n_ls, z, *rest):
    log_s, log_det = rest[:n_ls], rest[n_ls:]
    lib = _lib.load()
    z32 = z.float().contiguous()
    ls32 = [t.float().contiguous() for t in log_s]
    ptrs = (C.c_void_p * n_ls)(*[t.data_ptr() for t in ls32])
    sizes = (C.c_int64 * n_ls)(*[t.numel() for t in ls32])
    out = torch.empty((), dtype=torch.float32, device=z.device)
    ws = torch.empty(16, dtype=torch.uint8, device=z.device)
    stream = torch.cuda.current_stream(z.device).cuda_stream
    # log_det_W stays on the device (one small stack): reading the scalars on the host would synchronise the stream
    # between forward and backward in every training step
    ld = torch.stack([x.detach().reshape(()).to(device=z.device, dtype=torch.float32) for x in log_det]).contiguous()
    _lib.check(lib.wg_loss_dev(z32.data_ptr(), z32.numel(), ptrs, sizes, n_ls, ld.data_ptr(), float(sigma), out.data_ptr(),
                               ws.data_ptr(), ws.numel(), C.c_void_p(stream)))
    ctx.save_for_backward(z)
    ctx.sigma, ctx.n_ls = float(sigma), n_ls
    ctx.meta = [(t.shape, t.dtype) for t in log_s] + [(t.shape, t.dtype) for t in log_det]
    return out

  @staticmethod
  def backward(ctx, g):
    (z,) = ctx.saved_tensors
    n = z.numel()
    gz = z * (g / (ctx.sigma * ctx.sigma * n)).to(z.dtype)
    rest = [(-g / n).to(dt).expand(shape) for shape, dt in ctx.meta]
    return (None, None, gz, *rest)


class WaveGlowLoss(nn.Module):
  """src/waveglow/train.py:26-45: NLL of the flow, computed by the HIP library (``wg_loss``) on the tuple that
  ``WaveGlow.forward`` returns.  ``y`` is ignored, as in the reference (train.py:31-32).  Differentiable: the
  gradient w.r.t. z / log_s / log_det_W is closed-form and feeds the library's backward pass."""

  def __init__(self, sigma: float = 1.0):
    super().__init__()
    self.sigma = sigma

  def forward(self, y_pred, y=None):
    z, log_s_list, log_det_W_list = y_pred
    if z.device.type != "cuda":
      raise _lib.WgError("WaveGlowLoss runs on the GPU library only")
    log_det = [t if torch.is_tensor(t) else torch.tensor(float(t), device=z.device) for t in log_det_W_list]
    return _LossFn.apply(self.sigma, len(log_s_list), z, *log_s_list, *log_det)
