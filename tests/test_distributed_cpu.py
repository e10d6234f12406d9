"""N>1 path on CPU (gloo, world_size 2): the utterance sharding and the barrier + max-over-ranks timing that
bench.py and the CLI use.  The data path has no collective, so the only distributed logic is partition + timing."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from waveglow_amd.sharding import shard_list


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _worker(rank, world, port, n_items, q):
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  items = list(range(n_items))
  mine = shard_list(items, rank, world)
  # stand-in for per-utterance work: a checksum that depends only on the item
  local = torch.tensor([sum((i * 2654435761) % 1000003 for i in mine), len(mine)], dtype=torch.int64)
  gathered = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
  dist.all_gather(gathered, local)
  # timing reduction used by bench.py: barrier on both sides, MAX over ranks
  dist.barrier()
  elapsed = torch.tensor([1.0 + rank], dtype=torch.float64)
  dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
  dist.barrier()
  if rank == 0:
    q.put(([g.tolist() for g in gathered], float(elapsed.item())))
  dist.destroy_process_group()


def test_two_rank_sharding_and_timing_reduction():
  world, n_items = 2, 37
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
  for p in procs:
    p.start()
  gathered, elapsed = q.get(timeout=120)
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  assert sum(g[1] for g in gathered) == n_items                       # every utterance exactly once
  assert sum(g[0] for g in gathered) == sum((i * 2654435761) % 1000003 for i in range(n_items))
  assert abs(gathered[0][1] - gathered[1][1]) <= 1                    # balanced
  assert elapsed == 2.0                                               # max over ranks


def _grad_worker(rank, world, port, q):
  from waveglow_amd.distributed import GradientAllReducer
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  torch.manual_seed(0)
  params = [torch.nn.Parameter(torch.zeros(s)) for s in ((7, 5), (300,), (2, 3, 4), (1000,), (1,))]
  g = torch.Generator().manual_seed(100 + rank)
  for i, p in enumerate(params):
    if not (i == 2 and rank == 1):                       # one rank has no gradient for one parameter
      p.grad = torch.randn(p.shape, generator=g)
  red = GradientAllReducer(params, bucket_bytes=1500)    # forces several buckets
  assert len(red.buckets) >= 3
  red.reduce()
  if rank == 0:
    q.put([p.grad.clone() for p in params])
  dist.destroy_process_group()


def test_two_rank_gradient_allreduce_averages_buckets():
  """The exchange step of the training path: bucketed all-reduce = mean over ranks of every gradient."""
  world = 2
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
  for p in procs:
    p.start()
  got = q.get(timeout=120)
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  shapes = ((7, 5), (300,), (2, 3, 4), (1000,), (1,))
  want = []
  for rank in range(world):
    g = torch.Generator().manual_seed(100 + rank)
    want.append([torch.randn(s, generator=g) if not (i == 2 and rank == 1) else None for i, s in enumerate(shapes)])
  for i in range(len(shapes)):
    a, b = want[0][i], want[1][i]
    ref = (a + (b if b is not None else torch.zeros_like(a))) / world
    assert torch.allclose(got[i], ref, atol=1e-6), i
