"""GPU parity: the HIP path (through the C ABI) against golden fixtures and the CPU oracle.

Tolerance (BASELINE.json north_star): RMS(audio_gpu - audio_ref_cpu_fp32) <= 1e-3 on identical
mel + noise + weights, for fp32 and fp16 I/O.  The kernels use fp16 MFMA operands with fp32
accumulation and an fp32 flow state.
"""
import numpy as np
import pytest
import torch

from _cases import Case, oracle_cfg_from_hp, rms
from waveglow_amd import synthetic
from waveglow_amd.hparams import HParams
from waveglow_amd.model import WaveGlow

pytestmark = pytest.mark.gpu

RMS_TOL = 1e-3


def build_model(hp, sd, device="cuda:0", normed=False):
  m = WaveGlow(hp)
  if normed:
    m.load_state_dict(synthetic.to_weightnorm_form(sd))
  else:
    m = WaveGlow.remove_weightnorm(m)
    m.load_state_dict(sd)
  return m.to(device).eval()


def gpu_infer(model, mel, z_init, z_early, sigma, dtype=torch.float32):
  dev = next(model.parameters()).device
  ze = [z_early[k].to(dev, dtype) for k in sorted(z_early, reverse=True)]
  with torch.no_grad():
    out = model.infer_with_noise(mel.to(dev, dtype), z_init.to(dev, dtype), ze, sigma)
  torch.cuda.synchronize()
  return out.float().cpu()


@pytest.mark.parametrize("name", ["c64", "c256", "c512"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_infer_golden(name, dtype):
  c = Case(name)
  model = build_model(c.hp, c.sd)
  out = gpu_infer(model, c.mel, c.z_init, c.z_early, c.sigma, dtype)
  err = rms(out - c.audio)
  print(f"{name} {dtype}: rms err {err:.3e} max {float((out - c.audio).abs().max()):.3e} signal rms {rms(c.audio):.3f}")
  assert out.shape == c.audio.shape
  assert torch.isfinite(out).all()
  assert err <= RMS_TOL


@pytest.mark.parametrize("name", ["c64"])
def test_infer_from_weightnorm_checkpoint(name):
  """686-key (parametrizations.weight.original0/1) checkpoint form loads and folds like the reference."""
  c = Case(name)
  model = build_model(c.hp, c.sd, normed=True)
  out = gpu_infer(model, c.mel, c.z_init, c.z_early, c.sigma)
  ref = torch.from_numpy(c.npz["audio_from_weightnorm_ckpt"])
  assert rms(out - ref) <= RMS_TOL


@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (2, 33), (1, 130)])
def test_infer_ragged_lengths_vs_oracle(B, T):
  """L = 32*T not a multiple of the 128-column tile; edge tiles, guard rows, batch > 1."""
  from oracle import torch_oracle as O
  hp = HParams(n_channels=64, n_layers=8, n_flows=4, n_early_every=2)
  sd = synthetic.make_state_dict(hp, seed=11)
  mel = synthetic.make_mel(B, T, seed=T)
  z_init, z_early = synthetic.make_noise(hp, B, 32 * T, seed=100 + T)
  with torch.no_grad():
    ref = O.infer_ref(sd, mel, z_init, z_early, 0.7, oracle_cfg_from_hp(hp))
  out = gpu_infer(build_model(hp, sd), mel, z_init, z_early, 0.7)
  assert rms(out - ref) <= RMS_TOL, rms(out - ref)


def test_infer_c256_vs_oracle_medium():
  from oracle import torch_oracle as O
  hp = HParams()
  sd = synthetic.make_state_dict(hp, seed=0)
  B, T = 2, 40
  mel = synthetic.make_mel(B, T)
  z_init, z_early = synthetic.make_noise(hp, B, 32 * T)
  torch.set_num_threads(16)
  with torch.no_grad():
    ref = O.infer_ref(sd, mel, z_init, z_early, 0.6, oracle_cfg_from_hp(hp))
  model = build_model(hp, sd)
  for dtype in (torch.float32, torch.float16):
    out = gpu_infer(model, mel, z_init, z_early, 0.6, dtype)
    err = rms(out - ref)
    print(f"c256 B{B} T{T} {dtype}: rms err {err:.3e} (signal {rms(ref):.3f})")
    assert err <= RMS_TOL


@pytest.mark.parametrize("name", ["c64", "c256"])
def test_forward_golden(name):
  c = Case(name)
  model = build_model(c.hp, c.sd)
  wav = torch.from_numpy(c.npz["fwd_audio_in"])
  with torch.no_grad():
    z, log_s, log_det = model((c.mel.cuda(), wav.cuda()))
  torch.cuda.synchronize()
  z_ref = torch.from_numpy(c.npz["fwd_z"])
  assert z.shape == z_ref.shape
  assert rms(z.cpu() - z_ref) <= 2e-3 * max(1.0, rms(z_ref)), rms(z.cpu() - z_ref)
  for k, ls in enumerate(log_s):
    ref = torch.from_numpy(c.npz[f"fwd_log_s_{k}"])
    assert ls.shape == ref.shape
    assert rms(ls.cpu() - ref) <= 1e-3, (k, rms(ls.cpu() - ref))
  ld = np.array([float(x) for x in log_det], dtype=np.float32)
  np.testing.assert_allclose(ld, c.npz["fwd_log_det"], atol=2e-3)


def test_flow_round_trip_full_size():
  """Size-independent property at BASELINE configs[1] size (B=16, T=864, 256 ch): the flow is invertible,
  forward(infer(z)) returns the injected noise: cat(sigma*z_early[4], sigma*z_early[8], sigma*z_init)."""
  hp = HParams()
  sd = synthetic.make_state_dict(hp, seed=0)
  model = build_model(hp, sd)
  B, T, sigma = 16, 864, 0.6
  dev = "cuda:0"
  g = torch.Generator(device=dev).manual_seed(7)
  mel = (torch.randn(B, 80, T, device=dev, generator=g) * 2 - 5).clamp_(-11.5, 2.0)
  L = 32 * T
  z_init = torch.randn(B, 4, L, device=dev, generator=g)
  z8 = torch.randn(B, 2, L, device=dev, generator=g)
  z4 = torch.randn(B, 2, L, device=dev, generator=g)
  with torch.no_grad():
    audio = model.infer_with_noise(mel, z_init, [z8, z4], sigma)
    assert audio.shape == (B, 256 * T) and torch.isfinite(audio).all()
    z, log_s, _ = model((mel, audio))
  want = sigma * torch.cat([z4, z8, z_init], 1)
  err = rms((z - want).cpu())
  print("round trip rms err", err)
  assert err <= 3e-3


def test_unsupported_config_is_an_error_not_a_fallback():
  hp = HParams(n_channels=16, n_layers=3, n_flows=4, n_early_every=2)
  m = WaveGlow(hp).cuda()
  from waveglow_amd._lib import WgError
  with pytest.raises(WgError):
    m.infer(torch.zeros(1, 80, 4, device="cuda"))


def test_cpu_tensor_is_an_error_not_a_fallback():
  from waveglow_amd._lib import WgError
  hp = HParams(n_channels=64, n_flows=4, n_early_every=2)
  m = WaveGlow(hp)
  with pytest.raises(WgError):
    m.infer(torch.zeros(1, 80, 4))
