"""Why did the CPU oracle take 88 s at T=500 on a GPU-box host and 7.2 s on 8 threads in the build container?
Prints the CPU budget of this process (affinity, cgroup quota) and times oracle.infer_ref at T=64/128 over thread counts."""
import json
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _cases import oracle_cfg_from_hp  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402
from waveglow_amd import synthetic  # noqa: E402
from waveglow_amd.hparams import HParams  # noqa: E402

info = {"cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "torch_threads_default": torch.get_num_threads(),
        "loadavg": os.getloadavg()}
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
          "/sys/fs/cgroup/cpuset.cpus.effective"):
  try:
    info[f] = open(f).read().strip()
  except Exception as e:
    info[f] = f"<{type(e).__name__}>"
info["OMP_NUM_THREADS"] = os.environ.get("OMP_NUM_THREADS")
print(json.dumps(info), flush=True)

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


for thr in [int(a) for a in sys.argv[1:]] or [4, 8, 16, 32, 64]:
  if thr > info["affinity"]:
    continue
  torch.set_num_threads(thr)
  run(8)
  res = {"threads": thr}
  for T in (64, 128):
    res[f"T{T}_s"] = round(min(run(T), run(T)), 3)
  print(json.dumps(res), flush=True)
