"""Utterance sharding for multi-GPU inference: one process per GPU, contiguous shards, no data-path collective."""
from __future__ import annotations

from typing import List, Sequence, Tuple, TypeVar

T = TypeVar("T")


def shard_range(n_items: int, rank: int, world_size: int) -> Tuple[int, int]:
  """[start, end) of rank's contiguous shard; sizes differ by at most one, earlier ranks take the remainder."""
  assert 0 <= rank < world_size and n_items >= 0
  q, r = divmod(n_items, world_size)
  start = rank * q + min(rank, r)
  return start, start + q + (1 if rank < r else 0)


def shard_list(items: Sequence[T], rank: int, world_size: int) -> List[T]:
  s, e = shard_range(len(items), rank, world_size)
  return list(items[s:e])


def equal_shard_list(items: Sequence[T], rank: int, world_size: int) -> List[T]:
  """Contiguous shard of exactly floor(n / world_size) items for every rank (the n % world_size items at the end are
  left out).  For data-parallel TRAINING: every rank must take the same number of optimiser steps per epoch, or the
  gradient all-reduces of different ranks pair up steps of different epochs and the longer rank blocks on an
  unmatched collective at the end."""
  assert 0 <= rank < world_size
  per = len(items) // world_size
  return list(items[rank * per:(rank + 1) * per])
