"""CPU oracle at configs[0] (T=500) run repeatedly in ONE process: wall / user / sys time and page faults per run.
If the first run is slow and sys-dominated while repeats are fast, the cost is first-touch page faults of freshly mapped
memory in this VM (every conv output above glibc's mmap threshold is a new mapping), not arithmetic."""
import json
import os
import resource
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

hp = HParams()
sd = synthetic.make_state_dict(hp, seed=0)
cfg = oracle_cfg_from_hp(hp)
torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else 16)
print(json.dumps({"threads": torch.get_num_threads(), "thp": open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip()
                  if os.path.exists("/sys/kernel/mm/transparent_hugepage/enabled") else None,
                  "MALLOC_env": {k: v for k, v in os.environ.items() if k.startswith("MALLOC")}}), flush=True)


def run(T):
  mel = synthetic.make_mel(1, T)
  z_init, z_early = synthetic.make_noise(hp, 1, 32 * T)
  r0, t0, w0 = resource.getrusage(resource.RUSAGE_SELF), os.times(), time.perf_counter()
  with torch.no_grad():
    O.infer_ref(sd, mel, z_init, z_early, 0.6, cfg)
  w1, t1, r1 = time.perf_counter(), os.times(), resource.getrusage(resource.RUSAGE_SELF)
  print(json.dumps({"T": T, "wall_s": round(w1 - w0, 2), "user_s": round(t1.user - t0.user, 2), "sys_s": round(t1.system - t0.system, 2),
                    "minflt": r1.ru_minflt - r0.ru_minflt, "majflt": r1.ru_majflt - r0.ru_majflt,
                    "nvcsw": r1.ru_nvcsw - r0.ru_nvcsw, "nivcsw": r1.ru_nivcsw - r0.ru_nivcsw}), flush=True)


run(8)
for T in (128, 256, 384, 500, 500, 500):
  run(T)
