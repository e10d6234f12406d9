#!/usr/bin/env python3
"""Headline benchmark: WaveGlow-256 inference throughput (audio samples/s at 22.05 kHz) on MI355X.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

One step = one ``WaveGlow.infer`` pass (model.py:223-274 of the reference: upsample, 12 flows x 8 WN layers,
noise draws included) over one batch of BASELINE.json configs[1]: LJS-v3-shaped 256-channel model (synthetic
weights, identical architecture/format), batch 16 of random 80x864 mels, fp16 I/O, sigma 0.6.  Inputs are
resident in HBM before the timed region.  Multi-GPU = independent utterance shards per rank (no collective on
the data path); per-GPU work is fixed => weak scaling.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP16_DENSE_TFLOPS = 2500.0   # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PF dense)


def wn_layer_macs_per_group_step(hp) -> float:
  """Algorithmic MACs of ALL WN-layer launches per group-timestep (SURVEY.md 8(d) terms that the wn_layer
  kernel covers: cond slice + dilated in_layer + res_skip of every layer, plus WN.end)."""
  C_, NS = hp.n_channels, hp.n_mel_channels * hp.n_group
  per_flow = NS * 2 * C_ * hp.n_layers + hp.n_layers * (C_ * 2 * C_ * hp.kernel_size) + \
      (hp.n_layers - 1) * (C_ * 2 * C_) + C_ * C_
  total, rem = 0.0, hp.n_group
  for k in range(hp.n_flows):
    if k % hp.n_early_every == 0 and k > 0:
      rem -= hp.n_early_size
    total += per_flow + C_ * rem   # end: C x 2h_k
  return total


def wn_layer_executed_macs_per_group_step(hp) -> float:
  """MACs the wn_layer launches actually EXECUTE per group-timestep, after the algebraic folds of DESIGN.md section 2
  (cond_layer o upsample: K 640 -> 320; WN.end folded into the skip rows: the skip half of res_skip and `end` become
  16 rows; in_layers[0] o start: 3 K-steps of 64 instead of 3C/64 in the first layer of every WN)."""
  C_, M = hp.n_channels, hp.n_mel_channels
  per_flow = 0.0
  for i in range(hp.n_layers):
    k1 = (3 * 64 if i == 0 else 3 * C_) + 4 * M
    per_flow += 2 * C_ * k1 + (C_ * C_ if i < hp.n_layers - 1 else 0) + 16 * C_
  return per_flow * hp.n_flows


def workload_label(args) -> str:
  """Which BASELINE.json config the arguments are (index into its `configs` list), or "custom"."""
  key = (args.channels, args.batch, args.frames, args.dtype)
  named = {(256, 1, 500, "fp32"): "configs[0]", (256, 16, 864, "fp16"): "configs[1]", (512, 64, 864, "fp16"): "configs[2]",
           (256, 32, 4000, "fp16"): "configs[4] (one GPU's shard of 32)"}
  return named.get(key, "custom")


def pmc_traffic_bytes(args):
  """HBM bytes per wn_layer launch from the committed PMC passes (profiles/pmc_traffic.json), or None when the
  bench shape differs from the profiled one.  bench.py cannot collect PMC counters itself."""
  path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
  try:
    d = json.load(open(path))
    w = d["workload"]
    if (w["channels"], w["batch"], w["frames"]) != (args.channels, args.batch, args.frames):
      return None
    return (d["fetch_correction"] * d["fetch_size_kib"] + d["write_size_kib"]) * 1024.0
  except Exception:
    return None


def cpu_baseline(hp, sd, seconds_hint: float = 20.0):
  """The CPU oracle (port of the reference's fp32 infer, verified bit-equal to the reference in the build
  container) timed on this host's cores on a bounded sample: configs[0] shape, mel [1,80,500]."""
  from oracle import torch_oracle as O
  from waveglow_amd import synthetic
  sys.path.insert(0, os.path.join(ROOT, "tests"))
  from _cases import oracle_cfg_from_hp
  # reference CLI uses every core (set_torch_thread_to_max, src/waveglow/utils.py:27-29); here: the cores this
  # process may actually run on, capped at the GPU box's per-GPU share of 16
  try:
    cores = len(os.sched_getaffinity(0))
  except AttributeError:
    cores = os.cpu_count() or 1
  cores = max(1, min(cores, 16))
  torch.set_num_threads(cores)
  cfg = oracle_cfg_from_hp(hp)

  def run(T):
    mel = synthetic.make_mel(1, T)
    z_init, z_early = synthetic.make_noise(hp, 1, 32 * T)
    t0 = time.perf_counter()
    with torch.no_grad():
      O.infer_ref(sd, mel, z_init, z_early, 0.6, cfg)
    return time.perf_counter() - t0

  run(10)                        # warm-up (thread pool, oneDNN primitive caches)
  # Run time grows much faster than linearly in T on the GPU box's host (T = 256 -> 500 measured 3 s -> 88 s: the
  # [4096, 32T] cond tensor falls out of cache), so the sample is bounded at T = 256 and doubling stops as soon as
  # the next run is not predicted to fit the budget.
  T, best, spent = 32, None, 0.0
  while True:
    dt = run(T)
    spent += dt
    best = dt
    if T >= 256 or dt * 3.0 + spent > seconds_hint:
      break
    T = min(256, T * 2)
  # spend the rest of a ~10 s budget on repeats of the final size and report the best of them
  reps = 1
  while spent + best < 0.5 * seconds_hint and reps < 8:
    dt = run(T)
    spent += dt
    best = min(best, dt)
    reps += 1
  return {"value": round(256 * T / best, 1), "unit": "samples/s", "cores": cores, "kind": "port",
          "sample": f"oracle/torch_oracle.infer_ref fp32, mel [1,80,{T}] (configs[0] is T=500), sigma 0.6, best of {reps} runs: "
                    f"{best:.2f} s ({spent:.1f} s of CPU work in all)"}


def train_roofline(hp, B, S, steps, ms, cnt, overlapped_ms_per_step=None):
  """Roofline block of the training line: the dominant kernel is wgrad_kernel (weight gradients: dW = G^T X over all
  columns, MFMA-bound).  Algorithmic FLOPs per step = 2 x columns x sum over the layers of the weight matrices' sizes
  (in_layers + cond_layer slice [2C x (3C + M8)], res rows [C x C] except in a flow's last layer, end x skip [8 x C]),
  plus the upsample filter per phase [M8 x 4M]; time = hipEvents around the launches (classes of wg_profile_read).
  The events are taken in a pass of its own with WG_TRAIN_SERIAL=1 (every launch on one stream): in the timed region
  the weight-gradient launches run on a low-priority stream beside the other streams' kernels, and the time between
  their events (`overlapped_ms_per_step`) measures how the CUs are shared, not the kernel."""
  C_, nl, nf, M8 = hp.n_channels, hp.n_layers, hp.n_flows, hp.n_mel_channels * 8
  cols = B * (S // hp.n_group)
  per_flow = nl * (2 * C_ * (3 * C_ + M8) + 8 * C_) + (nl - 1) * C_ * C_
  flops = 2.0 * cols * (nf * per_flow + M8 * 4 * hp.n_mel_channels)
  t_wgrad = ms[6] / steps * 1e-3
  achieved = flops / t_wgrad / 1e12 if t_wgrad > 0 else 0.0
  n = max(1, int(cnt[6]))
  return {"bound": "mfma", "kernel": "wgrad_kernel", "achieved": round(achieved, 2), "peak": PEAK_FP16_DENSE_TFLOPS,
          "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP16_DENSE_TFLOPS, 4), "traffic": None,
          "avg_launch_ms": round(ms[6] / n, 4), "launches_timed": int(cnt[6]), "algorithmic_flops_per_step": flops,
          "kernel_ms_per_step": {"wgrad": round(ms[6] / steps, 3)},
          "timing": "hipEvents around every launch in a separate pass with WG_TRAIN_SERIAL=1 (one stream)",
          "overlapped_ms_per_step": overlapped_ms_per_step}


def bench_train(args, rank, world, dev, dist):
  """BASELINE configs[3]: the training step of src/waveglow/train.py:190-199 (forward, WaveGlowLoss, backward, Adam),
  data-parallel: per-GPU batch fixed (weak scaling), gradients averaged by one bucketed all-reduce per step."""
  from waveglow_amd import synthetic
  from waveglow_amd.distributed import GradientAllReducer
  from waveglow_amd.hparams import HParams
  from waveglow_amd.model import WaveGlow, WaveGlowLoss
  hp = HParams(n_channels=args.channels)
  model = WaveGlow(hp)
  model.load_state_dict(synthetic.to_weightnorm_form(synthetic.make_state_dict(hp, seed=0)))
  model = model.to(dev).train()
  B = args.batch if args.batch != 16 else 32
  S = args.segment
  F_ = 1 + S // 256
  mel = synthetic.make_mel(B, F_, seed=77 + rank).to(dev)
  g = torch.Generator().manual_seed(5 + rank)
  wav = (torch.rand(B, S, generator=g) * 0.6 - 0.3).to(dev)
  crit = WaveGlowLoss(1.0)
  opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)     # train.py:58-66; fused as in waveglow_amd/training.py:load_optimizer
  from waveglow_amd.train import enable_data_parallel
  fused = enable_data_parallel(model)                     # all-reduce inside backward(), overlapped, when world > 1
  red = None if fused else GradientAllReducer(model.parameters())

  def step():
    opt.zero_grad(set_to_none=True)
    loss = crit(model((mel, wav)), None)
    loss.backward()
    if red is not None:
      red.reduce()
    opt.step()
    return loss

  for _ in range(args.warmup):
    loss = step()
  eng = model._engine
  eng.lib.wg_profile_enable(eng.handle, 1 << 6)  # hipEvents around the wgrad launches only (one pair per launch), on their stream
  torch.cuda.synchronize(dev)
  if dist is not None:
    dist.barrier()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    loss = step()
  torch.cuda.synchronize(dev)
  if dist is not None:
    dist.barrier()
  elapsed = time.perf_counter() - t0
  assert torch.isfinite(loss.detach()).all()
  ms = (C.c_double * 8)()
  cnt = (C.c_int64 * 8)()
  eng.lib.wg_profile_read(eng.handle, ms, cnt, 8)
  overlapped = round(ms[6] / args.steps, 3)
  # the kernel on its own: a few more steps with every launch on one stream (not part of `value`)
  prof_steps = min(3, args.steps)
  os.environ["WG_TRAIN_SERIAL"] = "1"
  step()
  torch.cuda.synchronize(dev)
  eng.lib.wg_profile_enable(eng.handle, 1 << 6)
  for _ in range(prof_steps):
    step()
  torch.cuda.synchronize(dev)
  eng.lib.wg_profile_read(eng.handle, ms, cnt, 8)
  eng.lib.wg_profile_enable(eng.handle, 0)
  del os.environ["WG_TRAIN_SERIAL"]
  if dist is not None:
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
  if rank == 0:
    samples = B * S * world
    flops = 3 * 2.0 * (81235408 / 8 if args.channels == 256 else 261355984 / 8) * samples   # fwd + 2x bwd, SURVEY 8d
    print(json.dumps({
      "metric": "training samples/sec (WaveGlow-256 train step: forward + loss + backward + all-reduce + Adam)",
      "value": round(samples * args.steps / elapsed, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
      "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
      "scaling": "weak", "vs_baseline": None,
      "dtype": "f16 MFMA operands / saved activations / gradient planes, f32 accumulate, f32 weights + Adam",
      "data": "synthetic (random mels + uniform audio, random-init weight-normed parameters)",
      "config": {"workload": f"configs[3]: {args.channels}ch train step, batch={B}/GPU x {S} samples, {F_} mel frames",
                 "parallelism": f"dp{world}, per-flow gradient all-reduce (RCCL) overlapped with backward"},
      "loss": float(loss.detach()),
      "algorithmic_TFLOP_per_s": round(flops * args.steps / elapsed / 1e12, 1),
      "roofline": train_roofline(hp, B, S, prof_steps, ms, cnt, overlapped)}), flush=True)
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=10)
  ap.add_argument("--warmup", type=int, default=3)
  ap.add_argument("--batch", type=int, default=16)
  ap.add_argument("--frames", type=int, default=864)
  ap.add_argument("--channels", type=int, default=256)
  ap.add_argument("--dtype", default="fp16", choices=["fp16", "fp32"])
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--workload", default="infer", choices=["infer", "train"],
                  help="infer: BASELINE configs[1] (the headline metric, default); train: configs[3], one optimiser step "
                       "(forward + loss + backward + gradient all-reduce + Adam) on batch 32 x 16000 samples per GPU")
  ap.add_argument("--segment", type=int, default=16000)
  args = ap.parse_args()

  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  world = int(os.environ.get("WORLD_SIZE", "1"))
  if world != args.gpus:
    if args.gpus != 1 or world != 1:
      raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
  torch.cuda.set_device(local_rank)
  dev = torch.device("cuda", local_rank)
  dist = None
  if world > 1 or os.environ.get("WG_BENCH_FORCE_DIST"):   # the env var exercises the RCCL path on one GPU
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=dev)

  from waveglow_amd import synthetic
  from waveglow_amd.hparams import HParams
  from waveglow_amd.model import WaveGlow

  if args.workload == "train":
    return bench_train(args, rank, world, dev, dist)

  hp = HParams(n_channels=args.channels)
  sd = synthetic.make_state_dict(hp, seed=0)
  model = WaveGlow.remove_weightnorm(WaveGlow(hp))
  model.load_state_dict(sd)
  dtype = torch.float16 if args.dtype == "fp16" else torch.float32
  model = model.to(dev).eval()
  B, T = args.batch, args.frames
  mel = synthetic.make_mel(B, T, seed=1234 + rank).to(dev, dtype)
  sigma = 0.6
  torch.manual_seed(4321 + rank)

  with torch.no_grad():
    for _ in range(args.warmup):
      audio = model.infer(mel, sigma=sigma)
    eng = model._engine
    eng.lib.wg_profile_enable(eng.handle, 1)
    torch.cuda.synchronize(dev)
    if dist is not None:
      dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
      audio = model.infer(mel, sigma=sigma)
    torch.cuda.synchronize(dev)
    if dist is not None:
      dist.barrier()
    elapsed = time.perf_counter() - t0
  assert torch.isfinite(audio).all()

  ms = (C.c_double * 4)()
  cnt = (C.c_int64 * 4)()
  eng.lib.wg_profile_read(eng.handle, ms, cnt, 4)
  eng.lib.wg_profile_enable(eng.handle, 0)

  if dist is not None:
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

  if rank == 0:
    samples_per_step = B * T * 256 * world
    value = samples_per_step * args.steps / elapsed
    n_wn = int(cnt[2])
    avg_wn_ms = ms[2] / max(1, n_wn)
    launches_per_step = hp.n_flows * hp.n_layers
    flops_per_launch = 2.0 * wn_layer_macs_per_group_step(hp) * (B * T * 32) / launches_per_step
    achieved = flops_per_launch / (avg_wn_ms * 1e-3) / 1e12 if n_wn else 0.0
    executed_per_launch = 2.0 * wn_layer_executed_macs_per_group_step(hp) * (B * T * 32) / launches_per_step
    executed = executed_per_launch / (avg_wn_ms * 1e-3) / 1e12 if n_wn else 0.0
    out = {
      "metric": "audio samples/sec/GPU (22.05 kHz) WaveGlow-256 infer; real-time factor",
      "value": round(value, 1),
      "unit": "samples/s",
      "n_gpus": world,
      "steps": args.steps,
      "warmup": args.warmup,
      "ms_per_step": round(elapsed / args.steps * 1e3, 3),
      "higher_is_better": True,
      "scaling": "weak",
      "vs_baseline": None,
      "dtype": "f16 MFMA operands, f32 accumulate/flow state, %s I/O" % args.dtype,
      "data": "synthetic (random 80xT log-mels, synthetic weights of the LJS-v3 256ch architecture)",
      "config": {"workload": f"{workload_label(args)}: {args.channels}ch WaveGlow.infer, batch={B}/GPU mels 80x{T}, "
                             f"{args.dtype} I/O, sigma=0.6",
                 "per_gpu_samples_per_step": B * T * 256, "parallelism": f"utterance-sharded x{world}, no collective"},
      "real_time_factor": round(value / 22050.0, 1),
      "samples_per_s_per_gpu": round(value / world, 1),
      "roofline": {"bound": "mfma", "kernel": "wn_layer_kernel", "achieved": round(achieved, 2),
                   "peak": PEAK_FP16_DENSE_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP16_DENSE_TFLOPS, 4),
                   "traffic": pmc_traffic_bytes(args), "avg_launch_ms": round(avg_wn_ms, 4), "launches_timed": n_wn,
                   "algorithmic_flops_per_launch": flops_per_launch,
                   # what the matrix cores actually execute after the weight folds (fewer MACs for the same result)
                   "executed_flops_per_launch": executed_per_launch, "executed_achieved": round(executed, 2),
                   "executed_frac": round(executed / PEAK_FP16_DENSE_TFLOPS, 4),
                   "kernel_ms_per_step": {"mel_pack": round(ms[0] / args.steps, 3), "flow_start": round(ms[1] / args.steps, 3),
                                          "wn_layer": round(ms[2] / args.steps, 3), "memset": round(ms[3] / args.steps, 3)}},
    }
    if world == 1 and not args.no_cpu_baseline:
      out["cpu_baseline"] = cpu_baseline(hp, sd)
    print(json.dumps(out), flush=True)
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
