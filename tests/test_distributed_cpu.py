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
    q.put([p.grad.numpy().copy() for p in params])      # by value (see _dp_worker)
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
    assert torch.allclose(torch.from_numpy(got[i]), ref, atol=1e-6), i


# ---------------------------------------------------------------------------------------------------------------------
# The DEFAULT data-parallel path of the training direction (waveglow_amd/train.py): flat gradient buffer with one
# contiguous region per flow, flow_backward_schedule = one all-reduce per flow right behind that flow's backward.
# The reference has no data parallelism (src/waveglow/utils.py:347-350 is a dead comment), so the contract is
# "mean over ranks == single-process gradient of the concatenated batch" (train.py:44 normalises by the local batch).
# ---------------------------------------------------------------------------------------------------------------------
_DP_GEOM = dict(Cc=64, nl=3, nf=4, M8=128)


def _toy_flow_gradient(k, X, n_out):
  """Stand-in for flow k's backward: gradient of mean_b 0.5*||W_k x_b||^2 w.r.t. W_k (a deterministic function of the
  LOCAL batch, linear in the per-sample terms -- like the real gradients)."""
  W = torch.linspace(-1.0, 1.0, n_out * X.shape[1], dtype=torch.float64).view(n_out, X.shape[1]) * (k + 1)
  Y = X.double() @ W.t()                       # [B, n_out]
  return (Y.t() @ X.double() / X.shape[0]).float().reshape(-1)   # [n_out * in]


def _fill_region(bufs, k, X):
  """Writes flow k's stand-in gradients THROUGH THE VIEWS the library writes through (not through regions[k]), so the
  test also pins that the views of a flow lie inside that flow's region."""
  views = (bufs.dw1[k], bufs.db1[k], bufs.dw2[k], bufs.db2[k], bufs.dwes[k], bufs.dstart[k], bufs.dout_init[k], bufs.dw1x1[k])
  n = sum(v.numel() for v in views)
  g = _toy_flow_gradient(k, X, 8)
  g = g.repeat((n + g.numel() - 1) // g.numel())[:n]
  off = 0
  for v in views:
    v.copy_(g[off:off + v.numel()].view(v.shape))
    off += v.numel()


def _dp_worker(rank, world, port, q):
  from waveglow_amd.train import GradBuffers, flow_backward_schedule
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  gen = torch.Generator().manual_seed(5)
  X_all = torch.randn(2 * world, 16, generator=gen)      # the concatenated batch; this rank holds rows [2r, 2r+2)
  X = X_all[2 * rank:2 * rank + 2]
  bufs = GradBuffers(device="cpu", **_DP_GEOM)
  order, sent = [], []
  orig_all_reduce = dist.all_reduce

  def counting_all_reduce(t, *a, **k):
    sent.append(t.numel())
    return orig_all_reduce(t, *a, **k)
  dist.all_reduce = counting_all_reduce

  def run_flow(k):
    order.append(k)
    _fill_region(bufs, k, X)
    if k == 0:
      bufs.tail.copy_(_toy_flow_gradient(99, X, 8).repeat(bufs.tail.numel() // 128 + 1)[:bufs.tail.numel()])
  flow_backward_schedule(_DP_GEOM["nf"], run_flow, bufs, dist.group.WORLD)
  dist.all_reduce = orig_all_reduce
  if rank == 0:
    q.put((order, sent, bufs.flat.numpy().copy()))     # by value: a tensor would travel as a shared-memory handle that dies with this process
  dist.destroy_process_group()


def test_two_rank_flow_schedule_equals_concatenated_batch_gradient():
  from waveglow_amd.train import GradBuffers
  world = 2
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
  for p in procs:
    p.start()
  order, sent, flat = q.get(timeout=120)
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  nf = _DP_GEOM["nf"]
  assert order == list(reversed(range(nf)))                     # last flow first: its gradients are final first
  ref = GradBuffers(device="cpu", **_DP_GEOM)
  assert sent == [ref.flow_stride] * nf + [ref.tail.numel()]    # ONE message per flow + one for the upsample tail
  # single process on the concatenated batch
  gen = torch.Generator().manual_seed(5)
  X_all = torch.randn(2 * world, 16, generator=gen)
  for k in range(nf):
    _fill_region(ref, k, X_all)
  ref.tail.copy_(_toy_flow_gradient(99, X_all, 8).repeat(ref.tail.numel() // 128 + 1)[:ref.tail.numel()])
  assert torch.allclose(torch.from_numpy(flat), ref.flat, rtol=1e-5, atol=1e-6)


def test_grad_buffers_layout():
  """Every view lies inside its flow's region, views do not overlap, and the regions + tail tile the flat buffer."""
  from waveglow_amd.train import GradBuffers
  b = GradBuffers(device="cpu", **_DP_GEOM)
  b.flat.zero_()                                                # the buffer is NOT zero-filled by the constructor
  nf, nl = _DP_GEOM["nf"], _DP_GEOM["nl"]
  assert sum(r.numel() for r in b.regions) + b.tail.numel() == b.flat.numel()
  assert b.rec % 4 == 0 and b.flow_stride % 4 == 0              # 16-byte aligned records (float4 stores in the library)
  marks = 0
  for k in range(nf):
    for v in (b.dw1[k], b.db1[k], b.dw2[k], b.db2[k], b.dwes[k], b.dstart[k], b.dout_init[k], b.dw1x1[k]):
      assert float(v.abs().sum()) == 0.0                        # not written yet: no overlap with earlier views
      v.fill_(1.0)
      marks += v.numel()
    assert float(b.regions[k].sum()) == b.flow_stride           # exactly flow k's region is full
    assert float(b.flat.sum()) == (k + 1) * b.flow_stride
  b.dwup.fill_(1.0)
  b.dbup.fill_(1.0)
  assert float(b.flat.sum()) == b.flat.numel() and marks == nf * b.flow_stride
  assert b.dw1.shape[:2] == (nf, nl)


# ---------------------------------------------------------------------------------------------------------------------
# train() under data parallelism: equal step counts on every rank (odd file count), checkpoints by rank 0 only.
# The step's compute is stubbed (tiny CPU model + loader); the loop, sharding, cadence and the collective are real.
# ---------------------------------------------------------------------------------------------------------------------
def _train_worker(rank, world, port, n_files, tmp, q):
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  from pathlib import Path
  from torch.utils.data import DataLoader, Dataset
  from waveglow_amd import training, train as train_mod

  class Toy(torch.nn.Module):
    def __init__(self):
      super().__init__()
      self.lin = torch.nn.Linear(4, 1)

    def forward(self, x):
      return self.lin(x[0])

  class Items(Dataset):
    def __init__(self, entries):
      self.entries = list(entries)

    def __len__(self):
      return len(self.entries)

    def __getitem__(self, i):
      v = float(self.entries[i].stem)
      return torch.full((4,), v), torch.zeros(1)

  class Crit(torch.nn.Module):
    def __init__(self, sigma=1.0):
      super().__init__()

    def forward(self, y_pred, y=None):
      return (y_pred ** 2).mean()

  def toy_model(hparams, state_dict, device):
    torch.manual_seed(0)
    m = Toy()
    if state_dict is not None:
      m.load_state_dict(state_dict)
    return m
  training.load_model = toy_model
  training.prepare_trainloader = lambda hp, ts, dev: DataLoader(Items(ts), batch_size=hp.batch_size, drop_last=True)
  training.prepare_valloader = lambda hp, vs, dev: DataLoader(Items(vs), batch_size=hp.batch_size)
  training.WaveGlowLoss = Crit
  train_mod.enable_data_parallel = lambda model, group=None, force=False: False    # -> GradientAllReducer (CPU tensors)
  entries = [training.Entry(str(i), f"{i}.wav", Path(f"/nonexistent/{i}.wav")) for i in range(n_files)]
  custom = {"epochs": "2", "iters_per_checkpoint": "2", "batch_size": "1"}
  ckp = Path(tmp) / "ck"
  losses = training.train(custom, None, entries, entries[:1], ckp, None, None, torch.device("cpu"))
  model_after = training.get_last_checkpoint(ckp)[0] if rank == 0 else None
  w = torch.cat([p.detach().flatten() for p in torch.load(model_after, weights_only=True)["state_dict"].values()]) \
      if rank == 0 else torch.zeros(5)
  gathered = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
  dist.all_gather(gathered, torch.tensor([len(losses)], dtype=torch.int64))
  if rank == 0:
    q.put(([int(g) for g in gathered], training.get_all_checkpoint_iterations(ckp), w.numpy().copy()))
  dist.destroy_process_group()


def test_two_rank_train_loop_equal_steps_with_odd_file_count(tmp_path):
  """7 files on 2 ranks, batch size 1: contiguous shards of sizes 4 and 3 would give 4 and 3 steps per epoch and the
  all-reduces would pair steps of different epochs (and hang at the end); equal shards give 3 + 3."""
  world, n_files = 2, 7
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_train_worker, args=(r, world, port, n_files, str(tmp_path), q)) for r in range(world)]
  for p in procs:
    p.start()
  steps, ckpts, _ = q.get(timeout=180)
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  assert steps == [6, 6]                                        # 2 epochs x 3 steps on both ranks
  assert ckpts == [1, 2, 3, 4, 6]                               # first, every 2nd, epoch ends, last -- written once
