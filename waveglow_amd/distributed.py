"""Data-parallel training: the ONE exchange step of the path: all-reduce of the parameter gradients.

One process per GPU (``torch.distributed``; backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests), replicated
parameters.  The reference's loss divides by the LOCAL ``B*8*L`` (train.py:44), so averaging the gradients over the
ranks equals the single-process gradient of the concatenated batch.  Gradients travel as a few large flat fp32
buckets (xGMI rings are per-link bound: few, large messages), launched asynchronously in reverse parameter order
(the order in which the backward pass finishes them) and awaited together.
"""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist


class GradientAllReducer:
  def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 64 << 20, group=None):
    self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
    self.group = group
    self.buckets: List[List[torch.nn.Parameter]] = []
    cur, size = [], 0
    for p in reversed(self.params):
      n = p.numel() * 4
      if cur and size + n > bucket_bytes:
        self.buckets.append(cur)
        cur, size = [], 0
      cur.append(p)
      size += n
    if cur:
      self.buckets.append(cur)
    self._flat: List[torch.Tensor] = []

  def _buffers(self) -> List[torch.Tensor]:
    if not self._flat:
      for b in self.buckets:
        self._flat.append(torch.empty(sum(p.numel() for p in b), dtype=torch.float32, device=b[0].device))
    return self._flat

  @torch.no_grad()
  def reduce(self, force: bool = False) -> None:
    """Average ``p.grad`` over the process group, in place.  A parameter without a gradient counts as zero.
    ``force`` runs the collective even in a one-rank group (tests of the RCCL path on a single GPU)."""
    if not dist.is_available() or not dist.is_initialized():
      return
    world = dist.get_world_size(self.group)
    if world == 1 and not force:
      return
    flats = self._buffers()
    views, works = [], []
    for b, flat in zip(self.buckets, flats):
      off, vs = 0, []
      for p in b:
        v = flat[off:off + p.numel()].view_as(p)
        off += p.numel()
        vs.append(v)
      srcs = [p.grad if p.grad is not None else torch.zeros_like(p) for p in b]
      torch._foreach_copy_(vs, [s.to(torch.float32) for s in srcs])
      works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
      views.append(vs)
    for b, flat, vs, w in zip(self.buckets, flats, views, works):
      w.wait()
      flat.mul_(1.0 / world)
      for p, v in zip(b, vs):
        if p.grad is None:
          p.grad = v.clone().to(p.dtype)
        else:
          p.grad.copy_(v)
