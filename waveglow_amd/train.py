"""Training direction: ``WaveGlow.forward`` under autograd on the HIP library.

Reference: src/waveglow/model.py:178-221 (forward), train.py:190-199 (``loss.backward()``; Adam step on the
weight-normed parameters).  Everything that touches a weight, an activation or a gradient runs in the library:

* ``wg_train_prepare``: the module's own parameter tensors (weight-normed (g, v) pairs or dense weights, native layouts)
  -> every operand of the step: weight norm, the ``WN.end`` x skip fold, channel permutations, gate pre-scale, MFMA
  fragment orders (csrc/train_prep.hip, pack_kernel);
* ``wg_train_forward`` / ``wg_train_backward``: upsample, 12 x (1x1 conv, start, 8 x (dilated conv + cond + gate, res,
  skip/end), coupling) with saved fp16 activations, and the whole backward (both dgrads of every layer, every weight
  gradient, coupling / 1x1 / start backward) -- one C call each way;
* ``wg_train_param_grads``: packed gradients -> one gradient per parameter (weight-norm backward, the fold's chain rule) in
  ONE flat buffer; autograd gets views of it.

This file is the binding: the autograd node (``_TrainFn``), buffer allocation, the data-parallel schedule.  What is left
to torch is ``logdet`` of the twelve 1x1 weights (model.py:63) and the optimiser.  ``pack_weights`` /
``wn_forward_fragments`` / ``plain_fragments`` / ``to_fragments`` are the library's packing written as torch ops -- the
checker of tests/test_gpu_train.py::test_prepare_matches_torch_packing and tests/test_host_cpu.py, not on the product path.

There is no fallback: CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import List, Tuple

import torch

from . import _lib


def pos_perm(n: int) -> torch.Tensor:
  """perm[P] = channel stored at position P (wg_common.h: pos_to_chan) for n channels (n % 32 == 0)."""
  P = torch.arange(n)
  blk, p = P // 32, P % 32
  h, g, i = p // 16, (p // 4) % 4, p % 4
  return blk * 32 + 8 * g + 4 * h + i


_PERM_CACHE: dict = {}


def _perms(Cc: int, M8: int, device) -> "_Perms":
  """Index tensors are built (and copied to the device) once per geometry, not once per step."""
  key = (Cc, M8, str(device))
  if key not in _PERM_CACHE:
    _PERM_CACHE[key] = _Perms(Cc, M8, device)
  return _PERM_CACHE[key]


class _Perms:
  def __init__(self, Cc: int, M8: int, device):
    pc = pos_perm(Cc)
    self.c = pc.to(device)
    self.c2 = torch.cat([pc, Cc + pc]).to(device)
    self.m8 = pos_perm(M8).to(device)
    self.k1 = torch.cat([pc, Cc + pc, 2 * Cc + pc, 3 * Cc + pos_perm(M8)]).to(device)
    self.r32 = pos_perm(32).to(device)
    for name in ("c", "c2", "m8", "k1"):
      setattr(self, "i" + name, torch.argsort(getattr(self, name)))       # inverse permutations (gradients)
    r = torch.arange(32)
    self.c2p = (16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3)).to(device)   # chan_to_pos inside a 32-block


# (tensor index in the packed tuple, dims that carry a channel-position permutation, which permutation)
_PERM_PLAN = ((0, ((1, "c2"), (2, "k1"))), (1, ((1, "c2"),)), (2, ((1, "c"), (2, "c"))), (3, ((1, "c"),)), (4, ((2, "c"),)),
              (5, ((1, "m8"),)), (6, ((0, "m8"),)), (7, ((2, "c"),)))


def to_pos_order(packed, pm: "_Perms", inverse: bool = False):
  """Natural channel order -> the kernels' position-major order (or back, for gradients).  Plain gathers, done
  outside autograd (inside the autograd node): an indexing op under autograd costs a zero-filled full-size tensor and
  an atomic scatter in its backward; the inverse permutation is just another gather."""
  out = list(packed)
  for idx, dims in _PERM_PLAN:
    t = out[idx]
    for dim, name in dims:
      t = t.index_select(dim, getattr(pm, ("i" if inverse else "") + name))
    out[idx] = t
  return out


def _dense_stack(mods) -> torch.Tensor:
  """Dense weights of same-shaped conv modules, stacked [n, out, in, k].  Weight-normed modules (the training case) are
  evaluated with ONE ``torch._weight_norm`` over their concatenated (g, v) -- what the parametrization computes per
  module (rows are normalised independently), in one kernel each way instead of one per module."""
  is_p = torch.nn.utils.parametrize.is_parametrized
  if len(mods) > 1 and all(is_p(m, "weight") for m in mods):
    g = torch.cat([m.parametrizations.weight.original0 for m in mods])
    v = torch.cat([m.parametrizations.weight.original1 for m in mods])
    return torch._weight_norm(v, g, 0).view(len(mods), v.shape[0] // len(mods), *v.shape[1:])
  return torch.stack([m.weight for m in mods])


def _shape_runs(sizes):
  """[(size, [flow indices])] for maximal runs of consecutive flows with equal size (flow order is kept by concatenating
  the runs)."""
  runs = []
  for k, r in enumerate(sizes):
    if runs and runs[-1][0] == r:
      runs[-1][1].append(k)
    else:
      runs.append((r, [k]))
  return runs


def pack_weights(model) -> Tuple[torch.Tensor, ...]:
  """Differentiable packing of the module's parameters into the stacked matrices of wg_train_weights, in NATURAL
  channel order (to_pos_order applies the kernels' permutation).

  Few, LARGE autograd nodes: modules of one shape are gathered across ALL flows and go through one weight-norm
  evaluation (``_dense_stack``), and everything per flow (W_end x W_skip fold, start, 1x1) is a batched op over the
  flow axis -- a per-module formulation costs ~2000 tiny kernels per step (~8 ms of the round-1 step), and every
  per-layer slice of a big tensor costs a zero-filled full-size gradient in its backward."""
  hp = model._hp
  Cc, nl, nf, M, M8 = hp.n_channels, hp.n_layers, model.n_flows, hp.n_mel_channels, hp.n_mel_channels * 8
  pad = torch.nn.functional.pad
  WN = model.WN
  w_in = _dense_stack([WN[k].in_layers[i] for k in range(nf) for i in range(nl)])              # [FL, 2C, C, 3]
  b_in = torch.stack([WN[k].in_layers[i].bias for k in range(nf) for i in range(nl)])          # [FL, 2C]
  w_cond = _dense_stack([WN[k].cond_layer for k in range(nf)]).squeeze(3).reshape(nf * nl, 2 * Cc, M8)
  b_cond = torch.stack([WN[k].cond_layer.bias for k in range(nf)]).reshape(nf * nl, 2 * Cc)
  last_w = _dense_stack([WN[k].res_skip_layers[nl - 1] for k in range(nf)]).squeeze(3)         # [nf, C, C]: all skip rows
  last_b = torch.stack([WN[k].res_skip_layers[nl - 1].bias for k in range(nf)])                # [nf, C]
  if nl > 1:
    rs_w = _dense_stack([WN[k].res_skip_layers[i] for k in range(nf) for i in range(nl - 1)]).squeeze(3)
    rs_w = rs_w.view(nf, nl - 1, 2 * Cc, Cc)
    rs_b = torch.stack([WN[k].res_skip_layers[i].bias for k in range(nf) for i in range(nl - 1)]).view(nf, nl - 1, 2 * Cc)
    w2 = pad(rs_w[:, :, :Cc], (0, 0, 0, 0, 0, 1)).reshape(nf * nl, Cc, Cc)      # model.py:131-134; the last layer has no res rows
    b2 = pad(rs_b[:, :, :Cc], (0, 0, 0, 1)).reshape(nf * nl, Cc)
    w_skips = torch.cat([rs_w[:, :, Cc:], last_w[:, None]], 1)                  # [nf, nl, C, C]   model.py:135-136
    b_skip_sum = rs_b[:, :, Cc:].sum(1) + last_b                                # [nf, C]
  else:
    w2 = torch.zeros_like(last_w)
    b2 = torch.zeros_like(last_b)
    w_skips, b_skip_sum = last_w[:, None], last_b
  # Per-flow small tensors (end, start, 1x1): flows of one shape (h_k changes at the early outputs) are stacked and padded as
  # ONE tensor per shape group -- a per-flow formulation costs ~25 tiny kernels per flow and step each way.
  runs = _shape_runs([WN[k].end.weight.shape[0] for k in range(nf)])
  # WN.end (not weight-normed, 2h_k rows) zero-padded to 8 rows: end(sum_i skip_i) for every flow in one batched GEMM
  w_end8 = torch.cat([pad(torch.stack([WN[k].end.weight.squeeze(2) for k in ks]), (0, 0, 0, 8 - r)) for r, ks in runs])
  b_end8 = torch.cat([pad(torch.stack([WN[k].end.bias for k in ks]), (0, 8 - r)) for r, ks in runs])
  wes = torch.matmul(w_end8[:, None], w_skips).reshape(nf * nl, 8, Cc)           # [nf, nl, 8, C]
  out_init = torch.bmm(w_end8, b_skip_sum[:, :, None]).squeeze(2) + b_end8       # [nf, 8]
  # start conv [C, h_k] (weight norm per shape group), zero-padded to 4 columns, + bias row -> [nf, 5, C]
  start5 = torch.cat([torch.cat([pad(_dense_stack([WN[k].start for k in ks]).squeeze(3), (0, 4 - r // 2)).transpose(1, 2),
                                 torch.stack([WN[k].start.bias for k in ks])[:, None, :]], 1) for r, ks in runs])
  w1x1 = torch.cat([pad(torch.stack([model.convinv[k].conv.weight.squeeze(2) for k in ks]), (0, 8 - r, 0, 8 - r))
                    for r, ks in _shape_runs([model.convinv[k].conv.weight.shape[0] for k in range(nf)])])
  FL = nf * nl
  w_in = w_in.permute(0, 1, 3, 2).reshape(FL, 2 * Cc, 3 * Cc)                    # K = tap-major
  w1 = torch.cat([w_in, w_cond], 2)                                              # [FL, 2C, 3C + M8]
  # "* 1.0": AddBackward hands the SAME gradient tensor to both operands and cat / stack backward only slice it, so
  # in_layers[i].bias.grad and cond_layer.bias.grad would become overlapping views of one buffer (AccumulateGrad
  # installs them without a copy) and the next in-place accumulation would count a gradient twice
  b1 = b_in + b_cond * 1.0
  up = model.upsample.weight                                                     # [M_in, M_out, 1024]
  wup = up.view(M, M, 4, 32, 8).permute(3, 1, 4, 2, 0).reshape(32, M8, 4, M)     # [p][(o,g)][j][i]
  wup = torch.nn.functional.pad(wup, (0, 128 - M)).reshape(32, M8, 512)
  bup = model.upsample.bias.repeat_interleave(8)
  return (w1.contiguous(), b1.contiguous(), w2.contiguous(), b2.contiguous(), wes.contiguous(), wup.contiguous(),
          bup.contiguous(), start5.contiguous(), out_init.contiguous(), w1x1.contiguous())


def to_fragments(mat: torch.Tensor, c2p: torch.Tensor = None) -> torch.Tensor:
  """[..., M, K] (pos,pos) fp16 -> MFMA-fragment order [..., K/64, M/32, 4, 64, 8] (wg_train.h: PGemmArgs::A):
  lane (r, h), element j of sub-step s of block b, K-step t = mat[32b + chan_to_pos(r)][64t + 32h + 8s + j]."""
  *lead, M, K = mat.shape
  if c2p is None:
    r = torch.arange(32)
    c2p = (16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3)).to(mat.device)
  v = mat.reshape(*lead, M // 32, 32, K // 64, 2, 4, 8).index_select(len(lead) + 1, c2p)   # [.., b, r, t, h, s, j]
  n = len(lead)
  return v.permute(*range(n), n + 2, n, n + 4, n + 3, n + 1, n + 5).contiguous()            # [.., t, b, s, h, r, j]


def _ptr(t: torch.Tensor) -> C.c_void_p:
  return C.c_void_p(t.data_ptr())


K_TANH_SCALE = 2.8853900817779268     # 2*log2(e): the gate evaluates tanh / sigmoid through exp2 (kernels.hip: gate_act)
K_SIGM_SCALE = -1.4426950408889634    # -log2(e)


def wn_forward_fragments(w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, wes: torch.Tensor, pm: "_Perms", NW: int):
  """Natural-order stacked matrices -> the forward kernel's operands (include/waveglow_amd.h: wg_train_weights a1, a1c,
  b1, a2, es): rows stay in natural channel order, K goes to position order, the gate pre-scale is folded into the
  GEMM-1 rows and bias, and everything is laid out in the WN-layer kernel's MFMA A-fragment order.
  w1 [FL, 2C, 3C + M8], b1 [FL, 2C], w2 [FL, C, C], wes [FL, 8, C] (fp32)."""
  FL, C2, K1 = w1.shape
  Cc = C2 // 2
  MB = Cc // (32 * NW)
  nK = K1 // 64
  scale = torch.cat([torch.full((Cc,), K_TANH_SCALE), torch.full((Cc,), K_SIGM_SCALE)]).to(w1)
  wk = (w1.index_select(2, pm.k1) * scale[None, :, None]).half()                   # K in position order, rows natural
  # rows (gate, w, mb, r), K (ks, u1, k2, hh, j)  ->  [ks, u1, w, (gate, mb) = mt, k2, (hh, r) = lane, j]
  a = wk.reshape(FL, 2, NW, MB, 32, nK, 2, 2, 2, 8).permute(0, 5, 6, 2, 1, 3, 7, 8, 4, 9).contiguous()
  a = a.reshape(FL, nK, 2 * NW * 2 * MB * 2 * 64 * 8)
  n_tap = 3 * Cc // 64
  a1 = a[:, :n_tap].contiguous()
  a1c = a[:, n_tap:].contiguous()
  b1s = (b1 * scale[None, :]).float().contiguous()
  w2k = w2.index_select(2, pm.c).half()                                            # [FL, C rows natural, C pos]
  a2 = w2k.reshape(FL, NW, MB, 32, Cc // 16, 2, 8).permute(0, 1, 2, 4, 5, 3, 6).contiguous()   # [FL, w, mb, k16, hh, r, j]
  wek = wes.index_select(2, pm.c)
  hi = wek.half()
  lo = (wek - hi.float()).half()
  rows16 = torch.cat([hi, lo], 1)                                                  # [FL, 16, C pos]
  es = rows16.reshape(FL, 16, Cc // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()  # [FL, s, l4, row, j]: lane = 16 l4 + row
  return a1, a1c, b1s, a2, es


def plain_fragments(mat: torch.Tensor, NW: int) -> torch.Tensor:
  """[FL, M, K] fp16 (rows natural, K already in the order of the planes it multiplies) -> the WN-layer kernel's A-fragment
  order for plain row blocks [FL, K/64, 2, NW, MB, 2, 64, 8] (include/waveglow_amd.h: wat, wbt)."""
  FL, M, K = mat.shape
  MB = M // (32 * NW)
  v = mat.reshape(FL, NW, MB, 32, K // 64, 2, 2, 2, 8)                      # rows (w, mb, r), K (ks, u1, k2, hh, j)
  return v.permute(0, 4, 5, 1, 2, 6, 7, 3, 8).contiguous()                  # [FL, ks, u1, w, mb, k2, hh, r, j]


def canonical_params(model, eng):
  """(tensors in the library's canonical order, weight_normed flag) for wg_train_prepare / wg_train_param_grads: the
  module's own parameter tensors, looked up by state_dict key (include/waveglow_amd.h: wg_train_param_name)."""
  wn = torch.nn.utils.parametrize.is_parametrized(model.WN[0].start, "weight")
  key = ("train_param_names", wn)
  names = eng.cache.get(key)
  if names is None:
    n = eng.lib.wg_train_param_count(eng.handle, int(wn))
    names = [eng.lib.wg_train_param_name(eng.handle, int(wn), i).decode() for i in range(n)]
    eng.cache[key] = names
  byname = dict(model.named_parameters())
  try:
    tensors = [byname[n] for n in names]
  except KeyError as e:
    raise _lib.WgError(f"parameter {e} missing: the training direction needs every WN module in the same form "
                       "(all weight-normed, or all dense after remove_weightnorm)")
  for n, t in zip(names, tensors):
    if t.dtype != torch.float32 or not t.is_contiguous() or t.device.type != "cuda":
      raise _lib.WgError(f"parameter {n}: the training direction takes contiguous float32 parameters on the GPU")
  return names, tensors, wn


class _Weights:
  """Device buffers + the ctypes struct handed to the library (kept alive between forward and backward).  Everything in
  them -- fp16 fragment tensors and the small fp32 vectors -- is filled by the library itself from the module's own
  parameter tensors (``wg_train_prepare``: weight norm, the W_end x W_skip fold, permutations, gate pre-scale, fragment
  orders).  ``pack_weights`` / ``wn_forward_fragments`` / ``plain_fragments`` / ``to_fragments`` above are the same
  computation written as torch ops: the tests hold the library to them."""

  def __init__(self, model, tensors, wn: bool, flow_c: List[int], eng, stream):
    hp = model._hp
    Cc, nf, nl = hp.n_channels, model.n_flows, hp.n_layers
    M8 = hp.n_mel_channels * 8
    dev = tensors[0].device
    FL = nf * nl
    K1 = 3 * Cc + M8
    h16 = lambda n: torch.empty(n, dtype=torch.float16, device=dev)
    f32 = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
    n_tap = 3 * Cc // 64
    self.a1, self.a1c = h16(FL * n_tap * 64 * 2 * Cc), h16(FL * (K1 // 64 - n_tap) * 64 * 2 * Cc)
    self.a2, self.es = h16(FL * Cc * Cc), h16(FL * 16 * Cc)
    self.wat, self.wbt = h16(FL * Cc * (Cc + 64)), h16(FL * Cc * 6 * Cc)
    self.wct, self.wup = h16(M8 * FL * 2 * Cc), h16(32 * M8 * 512)
    self.b1, self.b2, self.bup = f32(FL, 2 * Cc), f32(FL, Cc), f32(M8)
    # per-flow operands as rows of one buffer each
    small = f32(nf, Cc * 4 + Cc + 8 + 64)
    self._small = small
    self.wstart = [small[k, :Cc * (c // 2)] for k, c in enumerate(flow_c)]
    self.bstart = [small[k, 4 * Cc:5 * Cc] for k in range(nf)]
    self.out_init = [small[k, 5 * Cc:5 * Cc + 8] for k in range(nf)]
    self.w1x1 = [small[k, 5 * Cc + 8:5 * Cc + 8 + c * c] for k, c in enumerate(flow_c)]
    arr = lambda ts: (C.c_void_p * nf)(*[t.data_ptr() for t in ts])
    self._arrs = [arr(self.wstart), arr(self.bstart), arr(self.out_init), arr(self.w1x1)]
    self.struct = _lib.WgTrainWeights(_ptr(self.a1), _ptr(self.a1c), _ptr(self.b1), _ptr(self.a2), _ptr(self.b2), _ptr(self.es),
                                      _ptr(self.wat), _ptr(self.wbt), _ptr(self.wct), _ptr(self.wup), _ptr(self.bup),
                                      C.cast(self._arrs[0], C.c_void_p), C.cast(self._arrs[1], C.c_void_p),
                                      C.cast(self._arrs[2], C.c_void_p), C.cast(self._arrs[3], C.c_void_p))
    self.wn = int(wn)
    self.params = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    self.aux = torch.empty(eng.lib.wg_train_prepare_bytes(eng.handle), dtype=torch.uint8, device=dev)
    _lib.check(eng.lib.wg_train_prepare(eng.handle, self.params, self.wn, C.byref(self.struct), _ptr(self.aux),
                                        self.aux.numel(), C.c_void_p(stream)))


class _SlotGuard:
  """Returns a training workspace to the pool when its autograd graph goes away -- after backward(), or when the
  graph is dropped without one (a loss that is only logged): otherwise every such forward would pin another workspace."""

  _serial = 0

  def __init__(self, slot: dict):
    _SlotGuard._serial += 1
    self.slot, self.token = slot, _SlotGuard._serial
    slot["owner"] = self.token        # a number, not a reference: the guard must die with its graph

  def release(self) -> None:
    if self.slot.get("owner") == self.token:
      self.slot["owner"] = None
      self.slot["busy"] = False          # stream-ordered: the next forward's kernels queue behind pending work

  def __del__(self):
    try:
      self.release()
    except Exception:
      pass


def _ddp_group(model):
  """Process group over which ``backward`` averages gradients itself (``model.ddp_group``; set by
  ``enable_data_parallel``), or None: single process, or the caller reduces with GradientAllReducer."""
  return getattr(model, "ddp_group", None)


def enable_data_parallel(model, group=None, force: bool = False) -> bool:
  """Average gradients inside ``backward``, flow by flow, overlapped with the rest of the backward pass.  Needs an
  initialised ``torch.distributed`` (backend "nccl" = RCCL); returns False (and changes nothing) in a single process
  unless ``force`` (tests of the RCCL path on one GPU)."""
  import torch.distributed as dist
  if not (dist.is_available() and dist.is_initialized()):
    return False
  if dist.get_world_size(group) == 1 and not force:
    return False
  model.ddp_group = group if group is not None else dist.group.WORLD
  return True


class GradBuffers:
  """All gradients of one backward pass in ONE flat fp32 buffer, laid out so that everything that becomes final
  with flow k's backward is contiguous (few, large messages -- xGMI rings are per-link bound):

    [ flow 0 region | flow 1 region | ... | tail ]
    flow region = n_layers records (dw1 | db1 | dw2 | db2 | dwes of one layer) | dstart [5,C] | dout_init [8] | dw1x1 [8,8]
    tail        = dwup [32, M8, 512] | dbup [M8]

  The library writes through strided views (wg_train_grads.layer_stride / flow_stride); ``regions[k]`` is what a
  data-parallel run all-reduces right behind flow k's backward, ``tail`` goes last.  12 + 1 messages per step at the
  reference's 12 flows (25.3 MB per flow, 5.2 MB tail at 256 channels)."""

  def __init__(self, Cc: int, nl: int, nf: int, M8: int, device):
    K1 = 3 * Cc + M8
    sizes = (2 * Cc * K1, 2 * Cc, Cc * Cc, Cc, 8 * Cc)
    self.rec = sum(sizes)
    small = (5 * Cc, 8, 64)
    self.flow_stride = nl * self.rec + sum(small)
    n_tail = 32 * M8 * 512 + M8
    # Not zero-filled (352 MB per step at 256 channels): the library writes every entry except dw2 / db2 of the last
    # layer of each flow (no res rows there, model.py:106-110), which are cleared below.  WG_TRAIN_POISON_GRADS=1
    # (tests) starts from NaN instead, so an entry the library leaves out cannot go unnoticed.
    self.flat = torch.empty(nf * self.flow_stride + n_tail, dtype=torch.float32, device=device)
    if os.environ.get("WG_TRAIN_POISON_GRADS") == "1":
      self.flat.fill_(float("nan"))
    self.regions = [self.flat[k * self.flow_stride:(k + 1) * self.flow_stride] for k in range(nf)]
    self.tail = self.flat[nf * self.flow_stride:]
    st = lambda shape, strides, off: self.flat.as_strided(shape, strides, off)
    off, fs, rec = 0, self.flow_stride, self.rec
    self.dw1 = st((nf, nl, 2 * Cc, K1), (fs, rec, K1, 1), off); off += sizes[0]
    self.db1 = st((nf, nl, 2 * Cc), (fs, rec, 1), off); off += sizes[1]
    self.dw2 = st((nf, nl, Cc, Cc), (fs, rec, Cc, 1), off); off += sizes[2]
    self.db2 = st((nf, nl, Cc), (fs, rec, 1), off); off += sizes[3]
    self.dwes = st((nf, nl, 8, Cc), (fs, rec, Cc, 1), off)
    off = nl * rec
    self.dstart = st((nf, 5, Cc), (fs, Cc, 1), off); off += small[0]
    self.dout_init = st((nf, 8), (fs, 1), off); off += small[1]
    self.dw1x1 = st((nf, 8, 8), (fs, 8, 1), off)
    self.dwup = self.tail[:32 * M8 * 512].view(32, M8, 512)
    self.dbup = self.tail[32 * M8 * 512:]
    self.nf = nf
    self.dw2[:, nl - 1].zero_()
    self.db2[:, nl - 1].zero_()

  def struct(self):
    """(wg_train_grads, keep-alive list)."""
    arr = lambda t: (C.c_void_p * self.nf)(*[t[k].data_ptr() for k in range(self.nf)])
    keep = [arr(self.dstart), arr(self.dout_init), arr(self.dw1x1)]
    g = _lib.WgTrainGrads(_ptr(self.dw1), _ptr(self.db1), _ptr(self.dw2), _ptr(self.db2), _ptr(self.dwes),
                          _ptr(self.dwup), _ptr(self.dbup), C.cast(keep[0], C.c_void_p), C.cast(keep[1], C.c_void_p),
                          C.cast(keep[2], C.c_void_p), self.rec, self.flow_stride)
    return g, keep

  def packed_grads(self) -> list:
    """Gradients in the shapes of pack_weights' outputs.  The library already writes them in natural channel order
    (wg_train_grads); the per-layer views are strided, so merging the (flow, layer) axes makes them dense."""
    merge = lambda t: t.reshape(t.shape[0] * t.shape[1], *t.shape[2:])
    return [merge(self.dw1), merge(self.db1), merge(self.dw2), merge(self.db2), merge(self.dwes), self.dwup, self.dbup,
            self.dstart.contiguous(), self.dout_init.contiguous(), self.dw1x1.contiguous()]


def flow_backward_schedule(n_flows: int, run_flow, bufs, group=None) -> None:
  """The data-parallel backward schedule, independent of what computes the gradients (so that the CPU tests drive it
  with a stub): flows are processed last to first; as soon as ``run_flow(k)`` has queued flow k's backward its region of
  the flat gradient buffer is final on the stream and goes out as ONE asynchronous all-reduce (RCCL runs on its own
  stream, ordered after the work queued so far), overlapping the backward of flows k-1 .. 0; the tail (upsample
  gradients, final with flow 0) follows; then everything is awaited and divided by the world size -- the reference's
  loss is a mean over the LOCAL batch (train.py:44), so the mean over ranks is the gradient of the concatenated batch.
  ``bufs`` needs ``regions`` (one tensor per flow), ``tail`` and ``flat``.  ``group`` None: no exchange."""
  if group is None:
    for k in reversed(range(n_flows)):
      run_flow(k)
    return
  import torch.distributed as dist
  works = []
  for k in reversed(range(n_flows)):
    run_flow(k)
    works.append(dist.all_reduce(bufs.regions[k], op=dist.ReduceOp.SUM, group=group, async_op=True))
  works.append(dist.all_reduce(bufs.tail, op=dist.ReduceOp.SUM, group=group, async_op=True))
  for wk in works:
    wk.wait()
  world = dist.get_world_size(group)
  if world > 1:
    bufs.flat.mul_(1.0 / world)


class _TrainFn(torch.autograd.Function):
  """Inputs: the module's parameters in the library's canonical order (``canonical_params``); outputs (z, log_s...).
  backward() returns one gradient per parameter, each a view of ONE flat buffer the library fills."""

  @staticmethod
  def forward(ctx, model, mel, audio, scale, wn, *params):
    eng = model._get_engine(mel.device, need_weights=False)
    lib = eng.lib
    B, M, F_ = mel.shape
    S = audio.shape[1]
    L = S // model.n_group
    flow_c = model.flow_channels()
    if int(lib.wg_wn_waves(model._hp.n_channels)) <= 0:
      raise _lib.WgError(f"n_channels={model._hp.n_channels} unsupported (64, 128, 256, 512)")
    stream = torch.cuda.current_stream(mel.device).cuda_stream
    wts = _Weights(model, [p.detach() for p in params], wn, flow_c, eng, stream)
    z = torch.empty((B, model.n_group, L), dtype=torch.float32, device=mel.device)
    log_s = [torch.empty((B, c // 2, L), dtype=torch.float32, device=mel.device) for c in flow_c]
    nbytes = lib.wg_train_workspace_bytes(eng.handle, B, F_, S)
    if nbytes == 0:
      raise _lib.WgError(lib.wg_last_error().decode())
    slot, fresh = eng.train_workspace(nbytes, (B, F_, S, nbytes))             # held until this graph's backward has run
    ws = slot["ws"]
    ls = (C.c_void_p * len(log_s))(*[t.data_ptr() for t in log_s])
    _lib.check(lib.wg_train_forward(eng.handle, C.byref(wts.struct), _ptr(mel), _ptr(audio), _ptr(z), ls, B, F_, S,
                                    1 if fresh else 0, _ptr(ws), ws.numel(), C.c_void_p(stream)))
    ctx.model, ctx.wts, ctx.ws, ctx.dims, ctx.audio, ctx.guard = model, wts, ws, (B, F_, S), audio, _SlotGuard(slot)
    ctx.scale = float(scale) if scale else float(2.0 ** round(math.log2(z.numel())))
    ctx.shapes = [t.shape for t in params]
    return (z, *log_s)

  @staticmethod
  def backward(ctx, g_z, *g_log_s):
    model, wts = ctx.model, ctx.wts
    eng = model._engine
    lib = eng.lib
    if ctx.wts is None:
      raise _lib.WgError("the saved activations of this forward pass are gone: backward() already ran for it "
                         "(retain_graph is not supported by the training direction)")
    B, F_, S = ctx.dims
    dev = ctx.audio.device
    nf = model.n_flows
    hp = model._hp
    bufs = GradBuffers(hp.n_channels, hp.n_layers, nf, hp.n_mel_channels * 8, dev)
    gstruct, _keep = bufs.struct()
    gz = g_z.float().contiguous() if g_z is not None else None
    gls = [g.float().contiguous() if g is not None else None for g in g_log_s]
    gl_arr = (C.c_void_p * nf)(*[(g.data_ptr() if g is not None else None) for g in gls])
    stream = torch.cuda.current_stream(dev).cuda_stream
    gz_ptr = _ptr(gz) if gz is not None else None
    group = _ddp_group(model)
    if group is None:
      _lib.check(lib.wg_train_backward(eng.handle, C.byref(wts.struct), C.byref(gstruct), gz_ptr, gl_arr,
                                       C.c_float(ctx.scale), _ptr(ctx.audio), B, F_, S, _ptr(ctx.ws), ctx.ws.numel(),
                                       C.c_void_p(stream)))
    else:
      # Data parallel: the backward pass is cut at flow boundaries and every flow's gradients -- ONE contiguous region
      # of the flat buffer -- are all-reduced right behind it (flow_backward_schedule).  What follows (weight-norm
      # backward, the fold's chain rule: wg_train_param_grads) is linear in these gradients, so averaging here equals
      # averaging the parameter gradients (the logdet term of the 1x1 weights is identical on every rank).
      def run_flow(k):
        _lib.check(lib.wg_train_backward_flows(eng.handle, C.byref(wts.struct), C.byref(gstruct), gz_ptr, gl_arr,
                                               C.c_float(ctx.scale), _ptr(ctx.audio), B, F_, S, _ptr(ctx.ws),
                                               ctx.ws.numel(), k, k, C.c_void_p(stream)))
      flow_backward_schedule(nf, run_flow, bufs, group)
    ctx.guard.release()
    # Overflow of the fp16 gradient planes (the automatic scale 2^round(log2 N) assumes the reference's MEAN loss; a
    # loss with another normalisation needs model.grad_scale) or inf / nan inputs: every gradient tensor is checked,
    # on the device.  model.grad_finite is read by waveglow_amd.training.train() before the optimiser step;
    # WG_TRAIN_CHECK_FINITE=1 raises here (one host sync per step).
    # (one reduction: inf / nan anywhere makes the sum non-finite, and 8.6e7 finite fp32 values cannot overflow it)
    model.grad_finite = torch.isfinite(bufs.flat.sum())
    if os.environ.get("WG_TRAIN_CHECK_FINITE") == "1" and not bool(model.grad_finite):
      ctx.wts = None
      raise _lib.WgError(nonfinite_message(ctx.scale))
    # one gradient per parameter, views of one flat buffer in the canonical order
    sizes = [math.prod(sh) for sh in ctx.shapes]
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
    if os.environ.get("WG_TRAIN_POISON_GRADS") == "1":
      flat.fill_(float("nan"))
    _lib.check(lib.wg_train_param_grads(eng.handle, wts.params, wts.wn, C.byref(gstruct), _ptr(wts.aux), wts.aux.numel(),
                                        _ptr(flat), C.c_void_p(stream)))
    ctx.wts = None
    grads = [v.view(sh) for v, sh in zip(flat.split(sizes), ctx.shapes)]
    return (None, None, None, None, None, *grads)


def nonfinite_message(scale: float) -> str:
  return ("non-finite gradients: the fp16 gradient planes overflowed (or the inputs held inf / nan).  The automatic loss "
          "scale assumes the reference's mean-normalised WaveGlowLoss (train.py:43-44); for a differently normalised loss "
          "set model.grad_scale (currently %s) so that scale * |dL/dz| stays below 65504" % (("%g" % scale) if scale else "automatic"))


def train_forward(model, mel: torch.Tensor, audio: torch.Tensor, grad_scale: float = 0.0):
  """(z, [log_s_k], [log_det_W_k]) with an autograd graph back to the module's parameters (model.py:178-221)."""
  if mel.device.type != "cuda":
    raise _lib.WgError("waveglow_amd runs on MI355X only: there is no CPU fallback")
  if mel.dtype != torch.float32 or audio.dtype != torch.float32:
    raise _lib.WgError("the training direction takes float32 mel / audio (reference: fp32 training)")
  B, M, F_ = mel.shape
  S = audio.shape[1]
  assert (F_ - 1) * 256 + 1024 >= S                    # model.py:187
  S = S - S % model.n_group                            # unfold drops the remainder (model.py:191,195)
  audio = audio[:, :S].contiguous()
  mel = mel.contiguous()
  eng = model._get_engine(mel.device, need_weights=False)
  if eng.width != eng.n_channels or eng.mel_width != eng.n_mel:
    raise _lib.WgError(f"n_channels={eng.n_channels}, n_mel_channels={eng.n_mel}: the training direction takes the kernel widths "
                       f"{eng.KERNEL_WIDTHS} and mel counts that are multiples of 16 only (inference and the no-grad forward "
                       "zero-pad the others)")
  _names, tensors, wn = canonical_params(model, eng)
  out = _TrainFn.apply(model, mel, audio, grad_scale, wn, *tensors)
  z, log_s = out[0], list(out[1:])
  L = S // model.n_group
  return z, log_s, _log_det_w(model, B * L)


def _log_det_w(model, n: int) -> List[torch.Tensor]:
  """``batch_size * n_of_groups * torch.logdet(W)`` of every flow (model.py:63) as ONE batched LU: each c_k x c_k matrix
  sits in the top-left corner of an 8 x 8 identity (same pivots, det x 1), so the twelve flows cost the ~20 small
  kernels of one ``logdet`` (forward and backward) instead of twelve times that -- 1.5 ms of launch latency per step at
  config 4.  Returns the reference's list of 0-dim tensors (views of one vector)."""
  ws = [model.convinv[k].conv.weight.squeeze(2) for k in range(model.n_flows)]
  dev = ws[0].device
  ng = model.n_group
  key = (str(dev), tuple(w.shape[0] for w in ws))
  cache = getattr(model, "_logdet_pad", None)
  if cache is None or cache[0] != key:
    base = torch.eye(ng, device=dev).repeat(len(ws), 1, 1)
    idx = torch.cat([(k * ng * ng + torch.arange(c)[:, None] * ng + torch.arange(c)[None, :]).reshape(-1)
                     for k, c in enumerate(key[1])]).to(dev)
    cache = (key, base.reshape(-1), idx)
    model._logdet_pad = cache
  _, base, idx = cache
  padded = base.index_put((idx,), torch.cat([w.reshape(-1) for w in ws])).view(len(ws), ng, ng)
  return list((n * torch.logdet(padded)).unbind(0))
