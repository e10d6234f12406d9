"""Per-phase cycle stamps of the training gate GEMM (plane_gemm_kernel<EPI_GATE>): one forward at config-4 shapes,
then the s_memtime stamps of the LAST gate launch: prologue / K loop / epilogue cycles per workgroup."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from waveglow_amd import _lib, synthetic  # noqa: E402
from waveglow_amd.hparams import HParams  # noqa: E402
from waveglow_amd.model import WaveGlow  # noqa: E402

hp = HParams()
model = WaveGlow(hp)
model.load_state_dict(synthetic.to_weightnorm_form(synthetic.make_state_dict(hp, seed=0)))
model = model.cuda().train()
B, S = 32, 16000
mel = synthetic.make_mel(B, 1 + S // 256, seed=7).cuda()
wav = (torch.rand(B, S) * 0.6 - 0.3).cuda()
lib = _lib.load()
lib.wg_train_debug_stamps.argtypes = [C.c_void_p]
buf = torch.zeros(4096 * 4, dtype=torch.int64, device="cuda")
for it in range(2):
  if it == 1:
    lib.wg_train_debug_stamps(C.c_void_p(buf.data_ptr()))
  y = model((mel, wav))
  torch.cuda.synchronize()
lib.wg_train_debug_stamps(None)
st = buf.cpu().numpy().reshape(-1, 4)
st = st[st[:, 0] != 0]
d = np.diff(st, axis=1).astype(np.float64)
print("workgroups", len(st))
print("s_memtime ticks: prologue %.0f  kloop %.0f  epilogue %.0f  total %.0f (median)" % tuple(np.median(np.c_[d, d.sum(1)], axis=0)))
print("mean:            prologue %.0f  kloop %.0f  epilogue %.0f  total %.0f" % tuple(np.mean(np.c_[d, d.sum(1)], axis=0)))
t0 = st[:, 0].min()
print("launch span ticks", st[:, 3].max() - t0, " first-start spread", np.percentile(st[:, 0] - t0, [0, 25, 50, 75, 100]))
