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
  16 rows; in_layers[0] o start: ONE gathered K-step of 64 instead of 3C/64 in the first layer of every WN, plus a 16-deep step that rebuilds the residual input x_0 from the a0 plane)."""
  C_, M = hp.n_channels, hp.n_mel_channels
  per_flow = 0.0
  for i in range(hp.n_layers):
    k1 = (64 if i == 0 else 3 * C_) + 4 * M
    if i == 0:
      per_flow += 16 * C_                       # x_0 = W_start a0 + b_start on the matrix pipe (wn_res_a0)
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


def _cpu_budget():
  """(affinity cores, cgroup CPU quota in cores or None): what this process may run on."""
  try:
    aff = len(os.sched_getaffinity(0))
  except AttributeError:
    aff = os.cpu_count() or 1
  quota = None
  try:
    q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if q != "max":
      quota = float(q) / float(per)
  except Exception:
    try:
      q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
      per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
      if q > 0:
        quota = q / per
    except Exception:
      pass
  return aff, quota


_CPU_CHILD = r"""
import json, os, sys, time
import torch
root, T, threads = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from oracle import torch_oracle as O
from waveglow_amd import synthetic
from waveglow_amd.hparams import HParams
from _cases import oracle_cfg_from_hp
torch.set_num_threads(threads)
hp = HParams()
sd = synthetic.make_state_dict(hp, seed=0)
cfg = oracle_cfg_from_hp(hp)
def run(T):
  mel = synthetic.make_mel(1, T)
  z_init, z_early = synthetic.make_noise(hp, 1, 32 * T)
  t0 = time.perf_counter()
  with torch.no_grad():
    O.infer_ref(sd, mel, z_init, z_early, 0.6, cfg)
  return time.perf_counter() - t0
run(8)
print(json.dumps({"T": T, "seconds": run(T)}), flush=True)
"""


def cpu_baseline(hp, sd, seconds_hint: float = 30.0, t500_limit_s: float = 110.0):
  """The CPU oracle (port of the reference's fp32 infer, verified bit-equal to the reference in the build
  container) timed on this host's cores at BASELINE configs[0]: mel [1,80,500], sigma 0.6.

  The reference CLI uses every core it sees (set_torch_thread_to_max, src/waveglow/utils.py:27-29).  On a GPU box the
  affinity mask (256) is far wider than the container's CPU quota (16) and the oneDNN/OpenMP pool thrashes beyond it, so the
  thread count is PROBED (T = 64 at a few candidate counts; the fastest is kept).  The run time is NOT linear in T: on the
  GPU boxes' hosts T = 256 takes ~3 s and T = 500 ~87 s (in the build container 5 s and 9-11 s, with T = 384 at 21 s: a
  stride / allocation pathology of the CPU convolutions at some lengths, not arithmetic), so T = 500 runs ONCE in a child
  process under a time limit; if it does not finish, the bounded T = 256 figure is reported instead (said in `sample`)."""
  import subprocess
  from oracle import torch_oracle as O
  from waveglow_amd import synthetic
  sys.path.insert(0, os.path.join(ROOT, "tests"))
  from _cases import oracle_cfg_from_hp
  aff, quota = _cpu_budget()
  cap = aff if quota is None else max(1, min(aff, int(round(quota))))
  cfg = oracle_cfg_from_hp(hp)

  def run(T):
    mel = synthetic.make_mel(1, T)
    z_init, z_early = synthetic.make_noise(hp, 1, 32 * T)
    t0 = time.perf_counter()
    with torch.no_grad():
      O.infer_ref(sd, mel, z_init, z_early, 0.6, cfg)
    return time.perf_counter() - t0

  cands = sorted({c for c in (4, 8, 16, 32, cap) if 1 <= c <= max(cap, 1)} | {min(cap, 8)})
  probe, spent = {}, 0.0
  for c in cands:
    torch.set_num_threads(c)
    run(8)                        # warm-up (thread pool, oneDNN primitive caches)
    dt = min(run(64), run(64))
    probe[c] = round(dt, 3)
    spent += 2 * dt
    if spent > 0.4 * seconds_hint:
      break
  cores = min(probe, key=probe.get)
  torch.set_num_threads(cores)
  t256 = min(run(256), run(256)) if probe[cores] * 4 * 2 * 1.5 < seconds_hint else run(256)
  out = {"unit": "samples/s", "cores": cores, "kind": "port", "threads_probe_s_at_T64": probe, "affinity_cores": aff,
         "cgroup_quota_cores": quota, "t256_s": round(t256, 2), "t256_samples_per_s": round(256 * 256 / t256, 1)}
  t500 = None
  try:
    res = subprocess.run([sys.executable, "-c", _CPU_CHILD, ROOT, "500", str(cores)], capture_output=True, text=True,
                         timeout=t500_limit_s, env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
    for ln in res.stdout.splitlines():
      if ln.startswith("{"):
        t500 = float(json.loads(ln)["seconds"])
  except subprocess.TimeoutExpired:
    t500 = None
  if t500 is not None:
    out.update({"value": round(256 * 500 / t500, 1), "t500_s": round(t500, 2),
                "sample": f"oracle/torch_oracle.infer_ref fp32, mel [1,80,500] = configs[0], sigma 0.6, one run in a child process: "
                          f"{t500:.2f} s (T = 256: {t256:.2f} s = {256 * 256 / t256:.0f} samples/s; the run time is not linear in T on "
                          f"this host, see cpu_baseline's docstring)"})
  else:
    out.update({"value": round(256 * 256 / t256, 1), "t500_s": None,
                "sample": f"oracle/torch_oracle.infer_ref fp32, mel [1,80,256] (configs[0] is T=500: did not finish in "
                          f"{t500_limit_s:.0f} s on this host), sigma 0.6: {t256:.2f} s"})
  return out


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


def run_train(channels, B, S, steps, warmup, rank, world, dev, dist):
  """BASELINE configs[3]: the training step of src/waveglow/train.py:190-199 (forward, WaveGlowLoss, backward, Adam),
  data-parallel: per-GPU batch fixed (weak scaling), gradients averaged by one all-reduce per flow inside backward().
  Returns the result dict (every rank; the times are the max over ranks)."""
  from waveglow_amd import synthetic
  from waveglow_amd.distributed import GradientAllReducer
  from waveglow_amd.hparams import HParams
  from waveglow_amd.model import WaveGlow, WaveGlowLoss
  hp = HParams(n_channels=channels)
  model = WaveGlow(hp)
  model.load_state_dict(synthetic.to_weightnorm_form(synthetic.make_state_dict(hp, seed=0)))
  model = model.to(dev).train()
  F_ = 1 + S // 256
  mel = synthetic.make_mel(B, F_, seed=77 + rank).to(dev)
  g = torch.Generator().manual_seed(5 + rank)
  wav = (torch.rand(B, S, generator=g) * 0.6 - 0.3).to(dev)
  crit = WaveGlowLoss(1.0)
  opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)     # train.py:58-66; fused as in waveglow_amd/training.py:load_optimizer
  from waveglow_amd.train import enable_data_parallel
  # all-reduce inside backward(), flow by flow, overlapped (world > 1; WG_BENCH_FORCE_DIST: also in a one-rank RCCL group)
  fused = enable_data_parallel(model, force=bool(os.environ.get("WG_BENCH_FORCE_DIST"))) if dist is not None else False
  red = None if (fused or dist is None) else GradientAllReducer(model.parameters())

  def step():
    opt.zero_grad(set_to_none=True)
    loss = crit(model((mel, wav)), None)
    loss.backward()
    if red is not None:
      red.reduce()
    opt.step()
    return loss

  for _ in range(warmup):
    loss = step()
  eng = model._engine
  torch.cuda.synchronize(dev)
  if dist is not None:
    dist.barrier()
  t0 = time.perf_counter()
  for _ in range(steps):
    loss = step()
  torch.cuda.synchronize(dev)
  if dist is not None:
    dist.barrier()
  elapsed = time.perf_counter() - t0
  assert torch.isfinite(loss.detach()).all()
  # Kernel timing is taken AFTER the timed region (profiling records ~200 hipEvents per step on the weight-gradient
  # stream): first as the step runs (streams overlapped), then with every launch on one stream (the kernel on its own).
  ms = (C.c_double * 8)()
  cnt = (C.c_int64 * 8)()
  prof_steps = min(3, steps)
  eng.lib.wg_profile_enable(eng.handle, 1 << 6)  # hipEvents around the wgrad launches only (one pair per launch), on their stream
  for _ in range(prof_steps):
    step()
  torch.cuda.synchronize(dev)
  eng.lib.wg_profile_read(eng.handle, ms, cnt, 8)
  overlapped = round(ms[6] / prof_steps, 3)
  os.environ["WG_TRAIN_SERIAL"] = "1"
  try:
    eng.lib.wg_profile_enable(eng.handle, 0)
    step()
    torch.cuda.synchronize(dev)
    eng.lib.wg_profile_enable(eng.handle, (1 << 4) | (1 << 5) | (1 << 6))
    for _ in range(prof_steps):
      step()
    torch.cuda.synchronize(dev)
    eng.lib.wg_profile_read(eng.handle, ms, cnt, 8)
    eng.lib.wg_profile_enable(eng.handle, 0)
  finally:
    del os.environ["WG_TRAIN_SERIAL"]
  if dist is not None:
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
  samples = B * S * world
  flops = 3 * 2.0 * (81235408 / 8 if channels == 256 else 261355984 / 8) * samples   # fwd + 2x bwd, SURVEY 8d
  roof = train_roofline(hp, B, S, prof_steps, ms, cnt, overlapped)
  roof["kernel_ms_per_step"].update({"wn_layer_forward": round(ms[4] / prof_steps, 3), "dgrad": round(ms[5] / prof_steps, 3)})
  out = {
    "metric": "training samples/sec (WaveGlow-256 train step: forward + loss + backward + all-reduce + Adam)",
    "value": round(samples * steps / elapsed, 1), "unit": "samples/s", "n_gpus": world, "steps": steps,
    "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True,
    "scaling": "weak", "vs_baseline": None,
    "dtype": "f16 MFMA operands / saved activations / gradient planes, f32 accumulate, f32 weights + Adam",
    "data": "synthetic (random mels + uniform audio, random-init weight-normed parameters)",
    "config": {"workload": f"configs[3]: {channels}ch train step, batch={B}/GPU x {S} samples, {F_} mel frames",
               "parallelism": f"dp{world}, per-flow gradient all-reduce (RCCL) overlapped with backward"},
    "loss": float(loss.detach()),
    "algorithmic_TFLOP_per_s": round(flops * steps / elapsed / 1e12, 1),
    "roofline": roof}
  del model, opt, eng
  torch.cuda.empty_cache()
  return out


def build_infer_model(channels, dev):
  from waveglow_amd import synthetic
  from waveglow_amd.hparams import HParams
  from waveglow_amd.model import WaveGlow
  hp = HParams(n_channels=channels)
  sd = synthetic.make_state_dict(hp, seed=0)
  model = WaveGlow.remove_weightnorm(WaveGlow(hp))
  model.load_state_dict(sd)
  return hp, sd, model.to(dev).eval()


def run_infer(hp, model, B, T, dtype_name, steps, warmup, rank, dev, dist, profile=True):
  """`steps` timed WaveGlow.infer calls (noise draws included) on one batch resident in HBM; returns
  (elapsed seconds: max over ranks, per-class kernel ms, launch counts).  `profile`: hipEvents around every launch
  inside the timed region (the headline: two event records per 0.5 ms launch are noise there); False: the events are
  taken in a pass of their own after the timed one (single-utterance latency: 110 launches of ~25 us per call)."""
  from waveglow_amd import synthetic
  dtype = torch.float16 if dtype_name == "fp16" else torch.float32
  mel = synthetic.make_mel(B, T, seed=1234 + rank).to(dev, dtype)
  sigma = 0.6
  torch.manual_seed(4321 + rank)
  ms = (C.c_double * 4)()
  cnt = (C.c_int64 * 4)()
  with torch.no_grad():
    for _ in range(warmup):
      audio = model.infer(mel, sigma=sigma)
    eng = model._engine
    if profile:
      eng.lib.wg_profile_enable(eng.handle, 1)
    torch.cuda.synchronize(dev)
    if dist is not None:
      dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
      audio = model.infer(mel, sigma=sigma)
    torch.cuda.synchronize(dev)
    if dist is not None:
      dist.barrier()
    elapsed = time.perf_counter() - t0
    if not profile:
      eng.lib.wg_profile_enable(eng.handle, 1)
      for _ in range(steps):
        audio = model.infer(mel, sigma=sigma)
      torch.cuda.synchronize(dev)
  assert torch.isfinite(audio).all()
  eng.lib.wg_profile_read(eng.handle, ms, cnt, 4)
  eng.lib.wg_profile_enable(eng.handle, 0)
  if dist is not None:
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
  return elapsed, list(ms), list(cnt)


def infer_roofline(hp, B, T, steps, ms, cnt, traffic=None):
  n_wn = int(cnt[2])
  avg_wn_ms = ms[2] / max(1, n_wn)
  launches_per_step = hp.n_flows * hp.n_layers
  flops_per_launch = 2.0 * wn_layer_macs_per_group_step(hp) * (B * T * 32) / launches_per_step
  achieved = flops_per_launch / (avg_wn_ms * 1e-3) / 1e12 if n_wn else 0.0
  executed_per_launch = 2.0 * wn_layer_executed_macs_per_group_step(hp) * (B * T * 32) / launches_per_step
  executed = executed_per_launch / (avg_wn_ms * 1e-3) / 1e12 if n_wn else 0.0
  return {"bound": "mfma", "kernel": "wn_layer_kernel", "achieved": round(achieved, 2),
          "peak": PEAK_FP16_DENSE_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP16_DENSE_TFLOPS, 4),
          "traffic": traffic, "avg_launch_ms": round(avg_wn_ms, 4), "launches_timed": n_wn,
          "algorithmic_flops_per_launch": flops_per_launch,
          # what the matrix cores actually execute after the weight folds (fewer MACs for the same result)
          "executed_flops_per_launch": executed_per_launch, "executed_achieved": round(executed, 2),
          "executed_frac": round(executed / PEAK_FP16_DENSE_TFLOPS, 4),
          "kernel_ms_per_step": {"mel_pack": round(ms[0] / steps, 3), "flow_start": round(ms[1] / steps, 3),
                                 "wn_layer": round(ms[2] / steps, 3), "memset": round(ms[3] / steps, 3)}}


def secondary_infer(label, hp, model, B, T, dtype_name, steps, warmup, rank, world, dev, dist):
  """One more BASELINE config, after the headline's timed region: the same measurement, condensed."""
  elapsed, ms, cnt = run_infer(hp, model, B, T, dtype_name, steps, warmup, rank, dev, dist, profile=False)
  roof = infer_roofline(hp, B, T, steps, ms, cnt)
  samples = B * T * 256 * world
  flops_per_sample = 2.0 * (81235408 if hp.n_channels == 256 else 261355984) / 8 if hp.n_channels in (256, 512) else None
  out = {"workload": label, "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3),
         "samples_per_s": round(samples * steps / elapsed, 1),
         "real_time_factor": round(samples * steps / elapsed / 22050.0, 1),
         "frac": roof["frac"], "executed_frac": roof["executed_frac"], "wn_layer_avg_launch_ms": roof["avg_launch_ms"],
         "wn_layer_ms_per_step": roof["kernel_ms_per_step"]["wn_layer"]}
  if flops_per_sample:
    out["whole_step_frac"] = round(flops_per_sample * samples * steps / elapsed / 1e12 / PEAK_FP16_DENSE_TFLOPS / world, 4)
  return out


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
  ap.add_argument("--no-secondary", action="store_true",
                  help="skip the other BASELINE configs (configs[0], [2], [3], [4]) that the default run measures after the "
                       "headline's timed region and reports under `secondary`")
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

  def finish():
    if dist is not None:
      dist.barrier()
      dist.destroy_process_group()

  if args.workload == "train":
    out = run_train(args.channels, args.batch if args.batch != 16 else 32, args.segment, args.steps, args.warmup,
                    rank, world, dev, dist)
    if rank == 0:
      print(json.dumps(out), flush=True)
    return finish()

  hp, sd, model = build_infer_model(args.channels, dev)
  B, T = args.batch, args.frames
  elapsed, ms, cnt = run_infer(hp, model, B, T, args.dtype, args.steps, args.warmup, rank, dev, dist)

  samples_per_step = B * T * 256 * world
  value = samples_per_step * args.steps / elapsed
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
    "roofline": infer_roofline(hp, B, T, args.steps, ms, cnt, pmc_traffic_bytes(args)),
  }

  # ---- the other BASELINE configs, each after the headline's timed region, in the SAME JSON line (`secondary`).
  # Every rank runs them (utterance shards: no collective; the training step all-reduces its gradients over RCCL).
  if not args.no_secondary and workload_label(args) == "configs[1]":
    sec = {}
    t_sec = time.perf_counter()
    try:
      sec["configs0"] = secondary_infer("configs[0]: 256ch, 1 mel 80x500, fp32 I/O (single-utterance latency, direct launches)",
                                        hp, model, 1, 500, "fp32", 20, 5, rank, world, dev, dist)
      sec["configs4_shard"] = secondary_infer("configs[4]: 256ch, one GPU's shard = 32 mels 80x4000, fp16 I/O",
                                              hp, model, 32, 4000, "fp16", 3, 1, rank, world, dev, dist)
      del model
      torch.cuda.empty_cache()
      hp5, _sd5, model5 = build_infer_model(512, dev)
      sec["configs2"] = secondary_infer("configs[2]: 512ch / 12 flows / 8 layers, 64 mels 80x864, fp16 I/O",
                                        hp5, model5, 64, 864, "fp16", 3, 1, rank, world, dev, dist)
      del model5, _sd5
      torch.cuda.empty_cache()
      tr = run_train(256, 32, 16000, 10, 3, rank, world, dev, dist)
      sec["train_configs3"] = {"workload": tr["config"]["workload"], "parallelism": tr["config"]["parallelism"],
                               "steps": tr["steps"], "warmup": tr["warmup"], "ms_per_step": tr["ms_per_step"],
                               "samples_per_s": tr["value"], "TFLOP_per_s": tr["algorithmic_TFLOP_per_s"],
                               "whole_step_frac": round(tr["algorithmic_TFLOP_per_s"] / PEAK_FP16_DENSE_TFLOPS / world, 4),
                               "loss": tr["loss"], "roofline": tr["roofline"]}
    except Exception as e:     # a secondary config must never take the headline line down with it
      sec["error"] = f"{type(e).__name__}: {e}"
    sec["wall_s"] = round(time.perf_counter() - t_sec, 1)
    out["secondary"] = sec

  if rank == 0:
    if world == 1 and not args.no_cpu_baseline:
      out["cpu_baseline"] = cpu_baseline(hp, sd)
    print(json.dumps(out), flush=True)
  finish()


if __name__ == "__main__":
  main()
