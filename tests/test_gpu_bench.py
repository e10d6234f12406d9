"""The commands the driver runs at round end, executed end to end as FRESH child processes on the GPU box:
``bench.py`` (headline configs[1] + the other BASELINE configs as `secondary`) and ``bench.py --workload train``, each
with ``WG_BENCH_FORCE_DIST=1`` so that the torch.distributed / RCCL code path of an N > 1 launch (process group,
barriers, max-over-ranks timing, per-flow gradient all-reduce inside backward) runs -- in a one-rank group, which is all
one GPU allows.  The numbers are not asserted (1 step); the contract of the JSON line is."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def _run_bench(extra):
  torch.cuda.empty_cache()                  # the child needs its own 22 GB of saved planes
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  port = s.getsockname()[1]
  s.close()
  env = dict(os.environ, WG_BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
             MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
  env.pop("WG_TRAIN_POISON_GRADS", None)
  res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline"] + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
  assert res.returncode == 0, res.stderr[-3000:]
  lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
  assert len(lines) == 1, res.stdout[-2000:]
  return json.loads(lines[0])


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline")


def test_bench_default_command_with_secondary_configs():
  out = _run_bench([])
  for key in CONTRACT:
    assert key in out, key
  assert out["n_gpus"] == 1 and out["steps"] == 1 and out["warmup"] == 1 and out["value"] > 0
  assert out["config"]["workload"].startswith("configs[1]")
  roof = out["roofline"]
  assert roof["bound"] == "mfma" and 0 < roof["frac"] < 1 and roof["launches_timed"] == 96
  sec = out["secondary"]
  assert "error" not in sec, sec
  for name in ("configs0", "configs2", "configs4_shard", "train_configs3"):
    assert sec[name]["ms_per_step"] > 0, name
  for name in ("configs0", "configs2", "configs4_shard"):
    assert 0 < sec[name]["frac"] < 1 and sec[name]["samples_per_s"] > 0
  tr = sec["train_configs3"]
  assert tr["workload"].startswith("configs[3]") and "batch=32/GPU x 16000" in tr["workload"]
  assert 0 < tr["roofline"]["frac"] < 1 and tr["roofline"]["kernel"] == "wgrad_kernel"
  assert tr["loss"] == tr["loss"]           # finite


def test_bench_train_command():
  out = _run_bench(["--workload", "train"])
  for key in CONTRACT:
    assert key in out, key
  assert out["config"]["workload"].startswith("configs[3]") and out["unit"] == "samples/s" and out["value"] > 0
  assert out["roofline"]["kernel"] == "wgrad_kernel" and out["roofline"]["launches_timed"] > 0
