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


@pytest.mark.parametrize("name", ["tiny", "c64", "c256", "c512"])
def test_infer_golden_fp32(name):
  """fp32 I/O against the committed output of the reference itself.  ("tiny": 16 channels -- a width the kernels are not
  instantiated for, run zero-padded at 64.)"""
  c = Case(name)
  model = build_model(c.hp, c.sd)
  out = gpu_infer(model, c.mel, c.z_init, c.z_early, c.sigma, torch.float32)
  err = rms(out - c.audio)
  print(f"{name} fp32: rms err {err:.3e} max {float((out - c.audio).abs().max()):.3e} signal rms {rms(c.audio):.3f}")
  assert out.shape == c.audio.shape
  assert torch.isfinite(out).all()
  assert err <= RMS_TOL


@pytest.mark.parametrize("name", ["c64", "c256", "c512"])
def test_infer_golden_case_fp16_io(name):
  """fp16 I/O: "identical mel + noise" means the fp16 tensors the kernel actually receives, so the CPU fp32
  oracle is evaluated on the fp16-rounded inputs of the golden case (the reference's own half path also draws its
  noise in fp16, model.py:234-237)."""
  from oracle import torch_oracle as O
  c = Case(name)
  mel16, z16 = c.mel.half().float(), c.z_init.half().float()
  ze16 = {k: v.half().float() for k, v in c.z_early.items()}
  with torch.no_grad():
    ref = O.infer_ref(c.sd, mel16, z16, ze16, c.sigma, c.oracle_cfg())
  model = build_model(c.hp, c.sd)
  out = gpu_infer(model, c.mel, c.z_init, c.z_early, c.sigma, torch.float16)
  err = rms(out - ref)
  print(f"{name} fp16 io: rms err {err:.3e} max {float((out - ref).abs().max()):.3e}")
  assert err <= RMS_TOL


@pytest.mark.parametrize("name", ["c64"])
def test_infer_from_weightnorm_checkpoint(name):
  """686-key (parametrizations.weight.original0/1) checkpoint form loads and folds like the reference."""
  c = Case(name)
  model = build_model(c.hp, c.sd, normed=True)
  out = gpu_infer(model, c.mel, c.z_init, c.z_early, c.sigma)
  ref = torch.from_numpy(c.npz["audio_from_weightnorm_ckpt"])
  assert rms(out - ref) <= RMS_TOL


@pytest.mark.parametrize("force_bn", ["128", "64"])
@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (2, 33), (1, 130)])
def test_infer_ragged_lengths_vs_oracle(B, T, force_bn, monkeypatch):
  """L = 32*T not a multiple of the 128-column tile; edge tiles, guard rows, batch > 1, several tiles per
  workgroup; both WN tile widths."""
  from oracle import torch_oracle as O
  monkeypatch.setenv("WG_FORCE_BN", force_bn)
  hp = HParams(n_channels=64, n_layers=8, n_flows=4, n_early_every=2)
  sd = synthetic.make_state_dict(hp, seed=11)
  mel = synthetic.make_mel(B, T, seed=T)
  z_init, z_early = synthetic.make_noise(hp, B, 32 * T, seed=100 + T)
  with torch.no_grad():
    ref = O.infer_ref(sd, mel, z_init, z_early, 0.7, oracle_cfg_from_hp(hp))
  out = gpu_infer(build_model(hp, sd), mel, z_init, z_early, 0.7)
  assert rms(out - ref) <= RMS_TOL, rms(out - ref)


@pytest.mark.parametrize("force_bn", ["128", "64"])
@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (2, 33)])
def test_infer_c256_edge_tiles_vs_oracle(B, T, force_bn, monkeypatch):
  """256 channels at lengths far below a tile: the 16x16x32 loop (128 columns) and the deep-prefetch 32x32x16 loop (64
  columns) on tiles that are mostly guard rows and padding, the first layer's gathered tap tile at the plane's edges."""
  from oracle import torch_oracle as O
  monkeypatch.setenv("WG_FORCE_BN", force_bn)
  hp = HParams(n_layers=5, n_flows=4, n_early_every=2)
  sd = synthetic.make_state_dict(hp, seed=21)
  mel = synthetic.make_mel(B, T, seed=T)
  z_init, z_early = synthetic.make_noise(hp, B, 32 * T, seed=200 + T)
  with torch.no_grad():
    ref = O.infer_ref(sd, mel, z_init, z_early, 0.7, oracle_cfg_from_hp(hp))
  out = gpu_infer(build_model(hp, sd), mel, z_init, z_early, 0.7)
  assert rms(out - ref) <= RMS_TOL, rms(out - ref)


@pytest.mark.parametrize("force_bn", ["128", "64"])
def test_infer_c256_vs_oracle_medium(force_bn, monkeypatch):
  """Both WN tile widths (128 columns = default, 64 = small-workload variant) against the oracle."""
  from oracle import torch_oracle as O
  monkeypatch.setenv("WG_FORCE_BN", force_bn)
  hp = HParams()
  sd = synthetic.make_state_dict(hp, seed=0)
  B, T = 2, 40
  mel = synthetic.make_mel(B, T)
  z_init, z_early = synthetic.make_noise(hp, B, 32 * T)
  torch.set_num_threads(16)
  model = build_model(hp, sd)
  for dtype in (torch.float32, torch.float16):
    rnd = (lambda t: t.to(dtype).float())
    with torch.no_grad():
      ref = O.infer_ref(sd, rnd(mel), rnd(z_init), {k: rnd(v) for k, v in z_early.items()}, 0.6, oracle_cfg_from_hp(hp))
    out = gpu_infer(model, mel, z_init, z_early, 0.6, dtype)
    err = rms(out - ref)
    print(f"c256 B{B} T{T} {dtype}: rms err {err:.3e} (signal {rms(ref):.3f})")
    assert err <= RMS_TOL


@pytest.mark.parametrize("channels", [128, 256])
@pytest.mark.parametrize("force_bn", ["128", "64"])
def test_first_layer_residual_from_a0_plane_matches_x0_planes(channels, force_bn, monkeypatch):
  """With the start fold the first layer of a WN rebuilds its residual input x_0 = W_start a0 + b_start from the a0
  plane (one MFMA step, hi + lo fp16 weights) and flow_kernel writes no x_0 planes; WG_NO_START_FOLD=1 (read at model
  creation) takes the stored fp16 x_0 like every other layer.  Both against the oracle, and against each other much
  closer than the bar."""
  from oracle import torch_oracle as O
  monkeypatch.setenv("WG_FORCE_BN", force_bn)
  hp = HParams(n_channels=channels, n_layers=4, n_flows=4, n_early_every=2)
  sd = synthetic.make_state_dict(hp, seed=5)
  B, T = 2, 37
  mel = synthetic.make_mel(B, T, seed=3)
  z_init, z_early = synthetic.make_noise(hp, B, 32 * T, seed=4)
  with torch.no_grad():
    ref = O.infer_ref(sd, mel, z_init, z_early, 0.8, oracle_cfg_from_hp(hp))
  out_fold = gpu_infer(build_model(hp, sd), mel, z_init, z_early, 0.8)
  monkeypatch.setenv("WG_NO_START_FOLD", "1")
  out_x0 = gpu_infer(build_model(hp, sd), mel, z_init, z_early, 0.8)
  monkeypatch.delenv("WG_NO_START_FOLD")
  assert rms(out_fold - ref) <= RMS_TOL and rms(out_x0 - ref) <= RMS_TOL, (rms(out_fold - ref), rms(out_x0 - ref))
  assert rms(out_fold - out_x0) <= 0.5 * RMS_TOL, rms(out_fold - out_x0)
  assert not torch.equal(out_fold, out_x0)          # the two paths are really different code


def test_layer_tile_widths_agree_c256(monkeypatch):
  """256 channels: the 128-column tile runs GEMM 1 on 16x16x32 MFMAs from its own weight packing, the 64-column tile on
  32x32x16 -- same math from different fragment layouts."""
  hp = HParams(n_flows=4, n_early_every=2)
  sd = synthetic.make_state_dict(hp, seed=9)
  B, T = 3, 50
  mel = synthetic.make_mel(B, T, seed=1)
  z_init, z_early = synthetic.make_noise(hp, B, 32 * T, seed=2)
  outs = {}
  for bn in ("128", "64"):
    monkeypatch.setenv("WG_FORCE_BN", bn)
    outs[bn] = gpu_infer(build_model(hp, sd), mel, z_init, z_early, 0.6)
  d = rms(outs["128"] - outs["64"])
  print(f"tile widths: rms diff {d:.3e} (signal {rms(outs['128']):.3f})")
  assert d <= 0.25 * RMS_TOL, d
  # ... in fact bit for bit on gfx950: both loops feed every accumulator its K values in the same order (bias, taps,
  # conditioning; k ascending inside a K-step) and the two MFMA shapes round alike, so a wrong fragment index in either
  # packing (api.cpp pack16 / cond_fold_kernel frag16) shows up here as a difference, not as a tolerance question
  assert torch.equal(outs["128"], outs["64"])


def test_weights_update_rebuilds_derived_state():
  """The reference caches W_inverse as a plain attribute and goes stale (model.py:52-58); here every packed layout
  incl. W^-1 is re-derived when parameters change -- also after .half()."""
  from oracle import torch_oracle as O
  hp = HParams(n_channels=64, n_layers=4, n_flows=4, n_early_every=2)
  sd1, sd2 = synthetic.make_state_dict(hp, seed=21), synthetic.make_state_dict(hp, seed=22)
  B, T = 1, 6
  mel = synthetic.make_mel(B, T)
  z_init, z_early = synthetic.make_noise(hp, B, 32 * T)
  model = build_model(hp, sd1)
  cfg = oracle_cfg_from_hp(hp)
  out1 = gpu_infer(model, mel, z_init, z_early, 0.9)
  model.load_state_dict(sd2)
  out2 = gpu_infer(model, mel, z_init, z_early, 0.9)
  with torch.no_grad():
    ref1 = O.infer_ref(sd1, mel, z_init, z_early, 0.9, cfg)
    ref2 = O.infer_ref(sd2, mel, z_init, z_early, 0.9, cfg)
  assert rms(out1 - ref1) <= RMS_TOL and rms(out2 - ref2) <= RMS_TOL
  model.half()
  out3 = gpu_infer(model, mel, z_init, z_early, 0.9, torch.float16)
  assert rms(out3 - ref2) <= 3e-3     # weights themselves were rounded to fp16 by .half()


@pytest.mark.parametrize("name", ["c64", "c256", "c512"])
def test_forward_golden(name):
  c = Case(name)
  model = build_model(c.hp, c.sd)
  wav = torch.from_numpy(c.npz["fwd_audio_in"])
  with torch.no_grad():
    z, log_s, log_det = model((c.mel.cuda(), wav.cuda()))
  torch.cuda.synchronize()
  z_ref = torch.from_numpy(c.npz["fwd_z"])
  assert z.shape == z_ref.shape
  assert rms(z.cpu() - z_ref) <= 1e-3 * max(1.0, rms(z_ref)), rms(z.cpu() - z_ref)
  for k, ls in enumerate(log_s):
    ref = torch.from_numpy(c.npz[f"fwd_log_s_{k}"])
    assert ls.shape == ref.shape
    assert rms(ls.cpu() - ref) <= 1e-3, (k, rms(ls.cpu() - ref))
  ld = np.array([float(x) for x in log_det], dtype=np.float32)
  np.testing.assert_allclose(ld, c.npz["fwd_log_det"], atol=2e-3)
  # WaveGlowLoss (train.py:31-45) on the device: on the reference's own forward outputs it must reproduce the
  # reference's loss (fp64 accumulation here vs fp32 there), and on ours it must stay within the flow's tolerance
  from waveglow_amd.model import WaveGlowLoss
  ref_out = (z_ref.cuda(), [torch.from_numpy(c.npz[f"fwd_log_s_{k}"]).cuda() for k in range(len(log_s))],
             [torch.tensor(float(v)) for v in c.npz["fwd_log_det"]])
  loss_ref_inputs = float(WaveGlowLoss(1.0)(ref_out, None))
  assert abs(loss_ref_inputs - float(c.npz["fwd_loss"])) <= 2e-6 * max(1.0, abs(float(c.npz["fwd_loss"])))
  loss_ours = float(WaveGlowLoss(1.0)((z, log_s, log_det), None))
  assert abs(loss_ours - float(c.npz["fwd_loss"])) <= 2e-3


# Round-trip bounds.  fp32 I/O: the only error is the kernels' own (fp16 MFMA operands).  fp16 I/O: the audio handed from
# infer to forward is ROUNDED to fp16 (relative 2^-11 of O(1) samples), an input perturbation of the forward pass that no
# kernel can undo; the bound is that perturbation's measured effect with margin, not a kernel tolerance.
ROUND_TRIP_TOL = {torch.float32: 1e-3, torch.float16: 1.5e-3}    # measured 2.6e-4 / 3.9e-4


def _round_trip(model, B, T, sigma, dtype, seed=7):
  dev = "cuda:0"
  g = torch.Generator(device=dev).manual_seed(seed)
  mel = (torch.randn(B, 80, T, device=dev, generator=g) * 2 - 5).clamp_(-11.5, 2.0).to(dtype)
  L = 32 * T
  z_init = torch.randn(B, 4, L, device=dev, generator=g).to(dtype)
  z8 = torch.randn(B, 2, L, device=dev, generator=g).to(dtype)
  z4 = torch.randn(B, 2, L, device=dev, generator=g).to(dtype)
  with torch.no_grad():
    audio = model.infer_with_noise(mel, z_init, [z8, z4], sigma)
    assert audio.shape == (B, 256 * T) and audio.dtype == dtype and torch.isfinite(audio).all()
    z, log_s, _ = model((mel, audio))
  want = sigma * torch.cat([z4, z8, z_init], 1).float()
  err = float((z.float() - want).double().pow(2).mean().sqrt())
  del z, log_s, audio
  return err


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_flow_round_trip_full_size(dtype):
  """Size-independent property at BASELINE configs[1] size (B=16, T=864, 256 ch), fp32 and fp16 I/O: the flow is
  invertible, forward(infer(z)) returns the injected noise: cat(sigma*z_early[4], sigma*z_early[8], sigma*z_init)."""
  hp = HParams()
  model = build_model(hp, synthetic.make_state_dict(hp, seed=0))
  err = _round_trip(model, 16, 864, 0.6, dtype)
  print(f"configs[1] round trip {dtype}: rms err {err:.3e}")
  assert err <= ROUND_TRIP_TOL[dtype]


def _summary_case(fixture, device="cuda:0"):
  """Inputs of a reference-generated SUMMARY fixture (tests/golden/make_golden_big.py): mel and the reference's own
  three noise draws, all fp16-representable (the generator rounds them before the reference sees them)."""
  import ast
  import os
  fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fixture))
  hp = HParams(**dict(ast.literal_eval(str(fx["hp_json"]))))
  sd = synthetic.make_state_dict(hp, seed=int(fx["weight_seed"]))    # crc32 vs the fixture: tests/test_oracle_golden.py
  T = int(fx["T"])
  mel = synthetic.make_mel(1, T, seed=int(fx["mel_seed"])).half()
  L = 32 * T
  torch.manual_seed(int(fx["noise_seed"]))
  z_init = torch.FloatTensor(1, 4, L).normal_().half()             # model.py:234-244
  z_early = []
  for k in reversed(range(hp.n_flows)):                            # model.py:260-271
    if k % hp.n_early_every == 0 and k > 0:
      z_early.append(torch.FloatTensor(1, hp.n_early_size, L).normal_().half())
  return fx, hp, sd, T, mel.to(device), z_init.to(device), [z.to(device) for z in z_early]


def _check_against_summary(fx, out):
  """out: fp32 cpu [n_samples].  RMS over the stored reference samples + whole-signal statistics."""
  assert out.numel() == int(fx["n_samples"])
  ref = np.concatenate([fx["first"], fx["strided"], fx["last"]])
  got = np.concatenate([out[:256].numpy(), out.numpy()[fx["strided_index"]], out[-256:].numpy()])
  err = float(np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2)))
  assert err <= RMS_TOL, err
  assert abs(float(out.double().pow(2).mean().sqrt()) - float(fx["rms"])) <= 1e-3
  assert abs(float(out.double().mean()) - float(fx["mean"])) <= 1e-3
  return err


def _batch_around(idx, B, T, lens, mel1, z1, ze1, seed):
  """A [B, 80, T] fp16 batch with the fixture utterance at `idx` and seeded random utterances elsewhere."""
  dev = mel1.device
  g = torch.Generator(device=dev).manual_seed(seed)
  L = 32 * T
  mel = (torch.randn(B, 80, T, device=dev, generator=g) * 2 - 5).clamp_(-11.5, 2.0).half()
  z_init = torch.randn(B, 4, L, device=dev, generator=g).half()
  z_early = [torch.randn(B, 2, L, device=dev, generator=g).half() for _ in ze1]
  mel[idx], z_init[idx] = mel1[0], z1[0]
  for z, zs in zip(z_early, ze1):
    z[idx] = zs[0]
  assert lens[idx] == T
  return mel, z_init, z_early


def test_infer_configs4_per_gpu_shard_fp16_against_reference_summary():
  """BASELINE configs[4], one GPU's shard: 256 channels, batch 32 of 80 x 4000 mels (47 s each), fp16 I/O, as ONE
  ragged batch.  Utterance 5 is the one the reference itself synthesised for the fixture (cfg5_summary.npz: its own
  infer() at [1,80,4000], sigma 0.6): <= 1e-3 RMS against the reference's samples; bit-identical to its batch-of-one
  call; and the whole shard round-trips through forward()."""
  fx, hp, sd, T, mel1, z1, ze1 = _summary_case("cfg5_summary.npz")
  assert T == 4000
  model = build_model(hp, sd)
  B, idx, sigma = 32, 5, float(fx["sigma"])
  lens = [T if b % 4 == 1 else T - 61 * b for b in range(B)]
  mel, z_init, z_early = _batch_around(idx, B, T, lens, mel1, z1, ze1, seed=11)
  with torch.no_grad():
    out = model.infer_with_noise(mel, z_init, z_early, sigma, frames=torch.tensor(lens, dtype=torch.int32))
    single = model.infer_with_noise(mel1, z1, ze1, sigma)
  torch.cuda.synchronize()
  assert out.dtype == torch.float16 and torch.isfinite(out).all()
  assert torch.equal(out[idx], single[0])
  for b in (0, 2, 31):
    assert float(out[b, 256 * lens[b]:].abs().max()) == 0.0 if lens[b] < T else True
  err = _check_against_summary(fx, out[idx].float().cpu())
  print(f"configs[4] utterance in a 32 x 80x4000 fp16 batch: rms err vs reference samples {err:.3e}")
  del out, single
  rt = _round_trip(model, B, T, sigma, torch.float16, seed=13)
  print(f"configs[4] shard round trip fp16: rms err {rt:.3e}")
  assert rt <= ROUND_TRIP_TOL[torch.float16]


def test_infer_configs2_full_size_fp16_against_reference_summary():
  """BASELINE configs[2] at full size: 512 channels, batch 64 of 80 x 864 mels, fp16 I/O.  Utterance 3 is the
  reference's own infer() of cfg3_summary.npz; same three checks as configs[4]."""
  fx, hp, sd, T, mel1, z1, ze1 = _summary_case("cfg3_summary.npz")
  assert T == 864 and hp.n_channels == 512
  model = build_model(hp, sd)
  B, idx, sigma = 64, 3, float(fx["sigma"])
  lens = [T if b % 4 == 3 else T - 7 * b for b in range(B)]
  mel, z_init, z_early = _batch_around(idx, B, T, lens, mel1, z1, ze1, seed=17)
  with torch.no_grad():
    out = model.infer_with_noise(mel, z_init, z_early, sigma, frames=torch.tensor(lens, dtype=torch.int32))
    single = model.infer_with_noise(mel1, z1, ze1, sigma)
  torch.cuda.synchronize()
  assert torch.isfinite(out).all()
  assert torch.equal(out[idx], single[0])
  err = _check_against_summary(fx, out[idx].float().cpu())
  print(f"configs[2] utterance in a 64 x 80x864 fp16 batch (512 ch): rms err vs reference samples {err:.3e}")
  del out, single
  rt = _round_trip(model, B, T, sigma, torch.float16, seed=19)
  print(f"configs[2] round trip fp16: rms err {rt:.3e}")
  assert rt <= ROUND_TRIP_TOL[torch.float16]


def test_infer_c256_weightnorm_checkpoint_with_nontrivial_g():
  """256-channel 686-key checkpoint whose g differs from ||v||: the reference folds it with its own remove_weightnorm
  (model.py:276-297) and synthesises c256_wn.npz; here the checkpoint is loaded in weight-normed form and folded by
  dense_state()."""
  import os
  fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c256_wn.npz"))
  hp = HParams()
  wn = synthetic.to_weightnorm_form(synthetic.make_state_dict(hp, seed=int(fx["weight_seed"])))
  gen = torch.Generator().manual_seed(int(fx["g_seed"]))
  for key in sorted(wn):
    if key.endswith("original0"):
      wn[key] = wn[key] * (0.5 + torch.rand(wn[key].shape, generator=gen))
  m = WaveGlow(hp)
  m.load_state_dict(wn)
  m = m.to("cuda:0").eval()
  T = int(fx["T"])
  mel = synthetic.make_mel(1, T, seed=int(fx["mel_seed"]))
  torch.manual_seed(int(fx["noise_seed"]))
  z_init = torch.FloatTensor(1, 4, 32 * T).normal_()
  z_early = {}
  for k in reversed(range(hp.n_flows)):
    if k % hp.n_early_every == 0 and k > 0:
      z_early[k] = torch.FloatTensor(1, hp.n_early_size, 32 * T).normal_()
  out = gpu_infer(m, mel, z_init, z_early, float(fx["sigma"]))
  err = rms(out - torch.from_numpy(fx["audio"]))
  print(f"c256 weight-normed checkpoint: rms err {err:.3e}")
  assert err <= RMS_TOL


@pytest.mark.parametrize("over", [dict(n_channels=640), dict(kernel_size=5), dict(n_layers=11)], ids=["c640", "k5", "l11"])
def test_unsupported_config_is_an_error_not_a_fallback(over):
  """Outside the envelope (n_channels <= 512, kernel_size 3, n_layers <= 10) the library says so; nothing else runs."""
  hp = HParams(n_flows=4, n_early_every=2, **{"n_channels": 64, "n_layers": 3, **over})
  m = WaveGlow(hp).cuda()
  from waveglow_amd._lib import WgError
  with pytest.raises(WgError):
    m.infer(torch.zeros(1, 80, 4, device="cuda"))


def test_training_direction_takes_kernel_widths_only():
  from waveglow_amd._lib import WgError
  hp = HParams(n_channels=96, n_layers=2, n_flows=2, n_early_every=1, n_early_size=2)
  m = WaveGlow(hp).cuda().train()
  with pytest.raises(WgError):
    m((torch.zeros(1, 80, 4, device="cuda"), torch.zeros(1, 1024, device="cuda")))


def test_cpu_tensor_is_an_error_not_a_fallback():
  from waveglow_amd._lib import WgError
  hp = HParams(n_channels=64, n_flows=4, n_early_every=2)
  m = WaveGlow(hp)
  with pytest.raises(WgError):
    m.infer(torch.zeros(1, 80, 4))


def test_synthesizer_and_cli_end_to_end(tmp_path):
  """Checkpoint file in the reference's format -> Synthesizer -> wav on disk through ``waveglow-cli synthesize``."""
  import numpy as np
  from scipy.io import wavfile
  from waveglow_amd.checkpoint import CheckpointWaveglow
  from waveglow_amd.synthesizer import Synthesizer
  from waveglow_amd import cli
  hp = HParams(n_channels=64, n_layers=4, n_flows=4, n_early_every=2)
  m = WaveGlow(hp)
  m.load_state_dict(synthetic.to_weightnorm_form(synthetic.make_state_dict(hp, seed=5)))
  ck_path = tmp_path / "10.pt"
  CheckpointWaveglow.from_instances(m, None, hp, 10).save(ck_path)
  ck = CheckpointWaveglow.load(ck_path, torch.device("cuda:0"))
  synth = Synthesizer(ck, device=torch.device("cuda:0"))
  mel = synthetic.make_mel(1, 12)
  r1 = synth.infer(mel, sigma=0.7, denoiser_strength=0.0005, seed=3)
  r2 = synth.infer(mel, sigma=0.7, denoiser_strength=0.0005, seed=3)
  assert r1.sampling_rate == 22050 and r1.wav.shape == (12 * 256,) and r1.wav_denoised.shape == (12 * 256,)
  assert np.array_equal(r1.wav, r2.wav)                    # seeded on every call (utils.py:221-229)
  assert r1.inference_duration_s > 0 and r1.denoising_duration_s > 0
  assert not np.array_equal(r1.wav, synth.infer(mel, sigma=0.7, seed=4).wav)
  mels = tmp_path / "mels" / "sub"
  mels.mkdir(parents=True)
  np.save(mels / "a.npy", mel[0].numpy())
  out = tmp_path / "out"
  rc = cli.main(["synthesize", str(ck_path), str(tmp_path / "mels"), "--sigma", "0.7", "--custom-seed", "3",
                 "--device", "cuda:0", "-out", str(out)])
  assert rc == 0
  rate, data = wavfile.read(out / "sub" / "a.wav")
  assert rate == 22050 and data.dtype == np.int16 and data.shape == (12 * 256,) and np.abs(data).max() == 32767
  # copy synthesis: wav -> mel (HIP front-end) -> wav  (waveglow-cli synthesize-wav, inference_wav.py:74-130)
  out2 = tmp_path / "out2"
  rc = cli.main(["synthesize-wav", str(ck_path), str(out), "--sigma", "0.7", "--custom-seed", "3",
                 "--device", "cuda:0", "-out", str(out2)])
  assert rc == 0
  rate2, data2 = wavfile.read(out2 / "sub" / "a.wav")
  # 3072 input samples -> 13 mel frames -> 13 * 256 output samples
  assert rate2 == 22050 and data2.dtype == np.int16 and data2.shape == (13 * 256,) and np.abs(data2).max() == 32767


def test_denoiser_stft_vs_numpy_oracle():
  """The HIP denoiser (exact-fp32 MFMA conv-STFT) against oracle/stft_oracle.py (numpy fp64 restatement of
  stft.py / denoiser.py; parity unpinned against the reference itself: its STFT module needs librosa)."""
  import ctypes as C
  import numpy as np
  from oracle import stft_oracle as S
  from waveglow_amd import _lib
  from waveglow_amd.denoiser import stft_bases
  lib = _lib.load()
  fwd, inv, wsq = stft_bases()
  h = C.c_void_p()
  _lib.check(lib.wg_stft_create(fwd.ctypes.data, inv.ctypes.data, wsq.ctypes.data, 1024, 256, 0, C.byref(h)))
  rng = np.random.default_rng(3)
  for B, T in ((2, 12), (1, 88), (3, 37)):
    x = (rng.standard_normal((B, 256 * T)) * 0.3).astype(np.float32)
    bias = np.abs(rng.standard_normal(513)).astype(np.float32) * 5.0
    strength = 0.1
    ref = S.denoise(x.astype(np.float64), bias.astype(np.float64), strength, *S.bases())
    re, im = S.transform(x.astype(np.float64), S.bases()[0])
    mag0_ref = np.sqrt(re ** 2 + im ** 2)[:, :, 0]
    xd, bd = torch.from_numpy(x).cuda(), torch.from_numpy(bias).cuda()
    out = torch.empty_like(xd)
    mag0 = torch.empty((B, 513), dtype=torch.float32, device="cuda")
    ws = torch.empty(lib.wg_stft_workspace_bytes(h, B, 256 * T), dtype=torch.uint8, device="cuda")
    _lib.check(lib.wg_stft_denoise(h, xd.data_ptr(), bd.data_ptr(), strength, out.data_ptr(), mag0.data_ptr(), B, 256 * T,
                                   ws.data_ptr(), ws.numel(), None))
    torch.cuda.synchronize()
    err = float(np.abs(out.cpu().numpy() - ref).max())
    print(f"denoise B{B} T{T}: max err {err:.2e}  mag0 err {float(np.abs(mag0.cpu().numpy() - mag0_ref).max()):.2e}")
    assert err <= 2e-5
    np.testing.assert_allclose(mag0.cpu().numpy(), mag0_ref, rtol=1e-4, atol=1e-4)
    # strength 0 / no bias -> identity (perfect reconstruction of the STFT pair)
    _lib.check(lib.wg_stft_denoise(h, xd.data_ptr(), None, 0.0, out.data_ptr(), None, B, 256 * T, ws.data_ptr(), ws.numel(), None))
    torch.cuda.synchronize()
    assert float((out - xd).abs().max()) <= 2e-5
  lib.wg_stft_destroy(h)


@pytest.mark.parametrize("channels,force_bn", [(64, None), (256, "128"), (256, "64"), (512, None)])
def test_ragged_batch_equals_batch_of_one_calls(channels, force_bn, monkeypatch):
  """wg_infer_ragged: utterances of different lengths padded into one batch give, each, bit-for-bit the audio of a
  batch-of-one call on their own frames (same injected noise), and zeros behind their own length."""
  if force_bn:
    monkeypatch.setenv("WG_FORCE_BN", force_bn)
  hp = HParams(n_channels=channels, n_layers=8, n_flows=4, n_early_every=2)
  sd = synthetic.make_state_dict(hp, seed=6)
  model = build_model(hp, sd)
  lens = [37, 5, 64, 21]
  Tm = max(lens)
  B = len(lens)
  mel = torch.full((B, 80, Tm), -11.5)
  zi0, ze0 = synthetic.make_noise(hp, 1, 32)
  z_init = torch.zeros(B, zi0.shape[1], 32 * Tm)
  z_e = {k: torch.zeros(B, v.shape[1], 32 * Tm) for k, v in ze0.items()}
  singles = []
  for b, T in enumerate(lens):
    m1 = synthetic.make_mel(1, T, seed=40 + b)
    zi, ze = synthetic.make_noise(hp, 1, 32 * T, seed=70 + b)
    mel[b, :, :T] = m1[0]
    z_init[b, :, :32 * T] = zi[0]
    for k in z_e:
      z_e[k][b, :, :32 * T] = ze[k][0]
    singles.append(gpu_infer(model, m1, zi, ze, 0.8))
    # garbage behind the utterance must not matter
    mel[b, :, T:] = 3.0
    z_init[b, :, 32 * T:] = 7.0
  dev = torch.device("cuda:0")
  with torch.no_grad():
    out = model.infer_with_noise(mel.to(dev), z_init.to(dev), [z_e[k].to(dev) for k in sorted(z_e, reverse=True)], 0.8,
                                 frames=torch.tensor(lens, dtype=torch.int32))
  torch.cuda.synchronize()
  out = out.cpu()
  for b, T in enumerate(lens):
    assert torch.equal(out[b, :256 * T], singles[b][0]), b
    assert float(out[b, 256 * T:].abs().max()) == 0.0 if T < Tm else True


def test_synthesizer_batch_equals_one_by_one(tmp_path):
  """Synthesizer.infer_batch / ``waveglow-cli synthesize --batch-size``: a ragged batch returns, per utterance, exactly
  what ``infer`` returns for it alone with the same seed (noise drawn the same way, denoiser per utterance)."""
  import numpy as np
  from scipy.io import wavfile
  from waveglow_amd import cli
  from waveglow_amd.checkpoint import CheckpointWaveglow
  from waveglow_amd.synthesizer import Synthesizer
  hp = HParams(n_channels=64, n_layers=4, n_flows=4, n_early_every=2)
  m = WaveGlow(hp)
  m.load_state_dict(synthetic.to_weightnorm_form(synthetic.make_state_dict(hp, seed=8)))
  ck_path = tmp_path / "3.pt"
  CheckpointWaveglow.from_instances(m, None, hp, 3).save(ck_path)
  synth = Synthesizer(CheckpointWaveglow.load(ck_path, torch.device("cuda:0")), device=torch.device("cuda:0"))
  mels = [synthetic.make_mel(1, T, seed=20 + T) for T in (9, 30, 17)]
  ones = [synth.infer(x, sigma=0.9, denoiser_strength=0.01, seed=11) for x in mels]
  batch = synth.infer_batch(mels, sigma=0.9, denoiser_strength=0.01, seed=11)
  assert len(batch) == 3
  for a, b in zip(ones, batch):
    assert np.array_equal(a.wav, b.wav) and np.array_equal(a.wav_denoised, b.wav_denoised)
  # CLI: batch size 2 over 3 files == batch size 1
  src = tmp_path / "mels"
  src.mkdir()
  for i, x in enumerate(mels):
    np.save(src / f"u{i}.npy", x[0].numpy())
  outs = []
  for bs in ("1", "2"):
    out = tmp_path / f"out{bs}"
    assert cli.main(["synthesize", str(ck_path), str(src), "--custom-seed", "5", "--batch-size", bs, "-out", str(out)]) == 0
    outs.append([wavfile.read(out / f"u{i}.wav")[1] for i in range(3)])
  for x, y in zip(*outs):
    assert np.array_equal(x, y)


def test_infer_graph_replay_matches_direct_launch():
  """hipGraph replay of wg_infer (single-utterance latency path) is bit-identical to the direct launch sequence --
  repeatedly (the workspace is re-zeroed by a kernel inside the graph), after the inputs change, and after the weights
  change (stale graphs are dropped together with the derived weights)."""
  hp = HParams(n_channels=64, n_layers=8, n_flows=4, n_early_every=2)
  sd = synthetic.make_state_dict(hp, seed=4)
  model = build_model(hp, sd)
  for seed in (1, 2, 3):
    mel = synthetic.make_mel(1, 20, seed=seed).cuda()
    z_init, z_early = synthetic.make_noise(hp, 1, 32 * 20, seed=10 + seed)
    ze = [z_early[k].cuda() for k in sorted(z_early, reverse=True)]
    with torch.no_grad():
      a = model.infer_with_noise(mel, z_init.cuda(), ze, 0.7)
      b = model.infer_with_noise(mel, z_init.cuda(), ze, 0.7, graph=True)
      c = model.infer_with_noise(mel, z_init.cuda(), ze, 0.7, graph=True)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(a, c)
  assert len(model._engine._graphs) == 1
  # a second, larger shape evicts the engine's cached workspace; graph A must keep ITS workspace alive (its device
  # pointer is baked into the captured launches) and still replay bit-identically, also after memory churn
  mel_b = synthetic.make_mel(1, 45, seed=9).cuda()
  zb, zeb = synthetic.make_noise(hp, 1, 32 * 45, seed=19)
  zeb = [zeb[k].cuda() for k in sorted(zeb, reverse=True)]
  with torch.no_grad():
    b_direct = model.infer_with_noise(mel_b, zb.cuda(), zeb, 0.7)
    b_graph = model.infer_with_noise(mel_b, zb.cuda(), zeb, 0.7, graph=True)
    churn = [torch.full((1 << 20,), 7.0, device="cuda") for _ in range(64)]     # re-use whatever memory was freed
    a_again = model.infer_with_noise(mel, z_init.cuda(), ze, 0.7, graph=True)
    b_again = model.infer_with_noise(mel_b, zb.cuda(), zeb, 0.7, graph=True)
  torch.cuda.synchronize()
  assert torch.equal(b_direct, b_graph) and torch.equal(b_direct, b_again) and torch.equal(a, a_again)
  assert all(float(c.min()) == 7.0 and float(c.max()) == 7.0 for c in churn)    # and nobody scribbled over live tensors
  assert len(model._engine._graphs) == 2
  with torch.no_grad():
    model.WN[0].end.bias.add_(0.01)
    a = model.infer_with_noise(mel, z_init.cuda(), ze, 0.7)
    b = model.infer_with_noise(mel, z_init.cuda(), ze, 0.7, graph=True)
  torch.cuda.synchronize()
  assert torch.equal(a, b)


def test_deep_prefetch_kernel_equals_one_step_ring(monkeypatch):
  """Small workloads at 256 channels / 80 mel channels run the two-step-deep weight-prefetch variant of wn_layer_kernel
  (hand-counted waits of its own, both for the regular and the first-layer K loops): same MFMA order, so it must agree BIT
  FOR BIT with the one-step ring (WG_DISABLE_DEEP=1, read per launch).  A wrong vmcnt or ring index shows up here first."""
  hp = HParams()                                     # 256 channels, 12 flows, 8 layers
  sd = synthetic.make_state_dict(hp, seed=6)
  model = build_model(hp, sd)
  for B, T in ((1, 40), (1, 131), (2, 37)):
    mel = synthetic.make_mel(B, T, seed=T).cuda()
    z_init, z_early = synthetic.make_noise(hp, B, 32 * T, seed=100 + T)
    ze = [z_early[k].cuda() for k in sorted(z_early, reverse=True)]
    with torch.no_grad():
      monkeypatch.delenv("WG_DISABLE_DEEP", raising=False)
      deep = model.infer_with_noise(mel, z_init.cuda(), ze, 0.6)
      monkeypatch.setenv("WG_DISABLE_DEEP", "1")
      ring = model.infer_with_noise(mel, z_init.cuda(), ze, 0.6)
    torch.cuda.synchronize()
    assert torch.isfinite(deep).all() and float(deep.abs().max()) > 1e-3
    assert torch.equal(deep, ring), (B, T, float((deep - ring).abs().max()))
  monkeypatch.delenv("WG_DISABLE_DEEP", raising=False)


def test_infer_configs0_shape_against_reference_summary():
  """BASELINE configs[0] (256 channels, mel [1,80,500] -> 128 000 samples, fp32): summary fixture written by the
  reference's own ``WaveGlow.infer`` (tests/golden/make_golden_cfg1.py).  The noise is the reference's: its three
  draws after ``torch.manual_seed`` are replayed here and injected."""
  import os
  fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg1_summary.npz"))
  hp = HParams()
  sd = synthetic.make_state_dict(hp, seed=int(fx["weight_seed"]))    # (crc32 vs the fixture: tests/test_oracle_golden.py)
  T, sigma = int(fx["T"]), float(fx["sigma"])
  mel = synthetic.make_mel(1, T, seed=int(fx["mel_seed"]))
  L = 32 * T
  torch.manual_seed(int(fx["noise_seed"]))
  z_init = torch.FloatTensor(1, 4, L).normal_()                   # model.py:234-244
  z_early = {}
  for k in reversed(range(hp.n_flows)):                           # model.py:260-271
    if k % hp.n_early_every == 0 and k > 0:
      z_early[k] = torch.FloatTensor(1, hp.n_early_size, L).normal_()
  model = build_model(hp, sd)
  out = gpu_infer(model, mel, z_init, z_early, sigma)[0]
  assert out.numel() == int(fx["n_samples"])
  ref = np.concatenate([fx["first"], fx["strided"], fx["last"]])
  got = np.concatenate([out[:256].numpy(), out.numpy()[fx["strided_index"]], out[-256:].numpy()])
  err = float(np.sqrt(np.mean((got - ref) ** 2)))
  print(f"configs[0]: rms err on 1536 reference samples {err:.3e}; rms {float(out.pow(2).mean().sqrt()):.5f} vs {float(fx['rms']):.5f}")
  assert err <= RMS_TOL
  assert abs(float(out.double().pow(2).mean().sqrt()) - float(fx["rms"])) <= 1e-3
  assert abs(float(out.double().mean()) - float(fx["mean"])) <= 1e-3
  assert abs(float(out.abs().max()) - float(fx["max_abs"])) <= 2e-2


@pytest.mark.parametrize("strength", [0.0005, 0.05])
def test_synthesizer_against_oracle_composition(tmp_path, strength):
  """Synthesizer.infer (src/waveglow/synthesizer.py:54-94) end to end against the composition of the CPU oracles:
  seed -> the three device draws of WaveGlow.infer (model.py:234-244, :260-271; replayed here after the same
  init_global_seeds) -> oracle infer_ref -> bias spectrum of infer(zeros[1,80,88], sigma 0) (denoiser.py:29-49) ->
  oracle spectral subtraction (denoiser.py:51-57).  The flow leg is pinned to the reference (torch_oracle); the
  denoiser leg is pinned only to oracle/stft_oracle.py (parity unpinned vs the reference: librosa absent)."""
  from oracle import stft_oracle as S
  from oracle import torch_oracle as O
  from waveglow_amd.checkpoint import CheckpointWaveglow
  from waveglow_amd.synthesizer import Synthesizer, init_global_seeds
  hp = HParams(n_channels=64, n_layers=4, n_flows=6, n_early_every=2)
  sd = synthetic.make_state_dict(hp, seed=15)
  m = WaveGlow(hp)
  m.load_state_dict(synthetic.to_weightnorm_form(sd))
  ck_path = tmp_path / "7.pt"
  CheckpointWaveglow.from_instances(m, None, hp, 7).save(ck_path)
  dev = torch.device("cuda:0")
  synth = Synthesizer(CheckpointWaveglow.load(ck_path, dev), device=dev)
  T, sigma, seed = 23, 0.8, 12
  mel = synthetic.make_mel(1, T, seed=3)
  res = synth.infer(mel, sigma=sigma, denoiser_strength=strength, seed=seed)
  # --- oracle composition
  cfg = oracle_cfg_from_hp(hp)
  L = 32 * T
  init_global_seeds(seed)
  z_init = torch.empty((1, 4, L), dtype=torch.float32, device=dev).normal_().cpu()
  z_early = {}
  for k in reversed(range(hp.n_flows)):
    if k % hp.n_early_every == 0 and k > 0:
      z_early[k] = torch.empty((1, hp.n_early_size, L), dtype=torch.float32, device=dev).normal_().cpu()
  with torch.no_grad():
    audio_ref = O.infer_ref(sd, mel, z_init, z_early, sigma, cfg)
    zb = torch.zeros(1, 4, 32 * 88)
    bias_audio = O.infer_ref(sd, torch.zeros(1, 80, 88), zb, {k: torch.zeros(1, 2, 32 * 88) for k in z_early}, 0.0, cfg)
  assert res.wav.dtype == np.float32 and res.wav.shape == (256 * T,)
  err = rms(torch.from_numpy(res.wav) - audio_ref[0])
  print(f"Synthesizer.wav vs oracle: rms err {err:.3e}")
  assert err <= RMS_TOL
  fwd, inv, wsq = S.bases()
  re, im = S.transform(bias_audio.double().numpy(), fwd)
  bias_mag = np.sqrt(re ** 2 + im ** 2)[0, :, 0]
  den_ref = S.denoise(audio_ref.double().numpy(), bias_mag, strength, fwd, inv, wsq)[0]
  err_d = float(np.sqrt(np.mean((res.wav_denoised.astype(np.float64) - den_ref) ** 2)))
  changed = float(np.sqrt(np.mean((den_ref - audio_ref[0].double().numpy()) ** 2)))
  print(f"Synthesizer.wav_denoised vs oracle composition: rms err {err_d:.3e} (the denoiser moved the signal by {changed:.3e})")
  assert res.wav_denoised.shape == (256 * T,)
  assert err_d <= RMS_TOL
  assert bool(res.was_overamplified) == bool(np.abs(res.wav).max() > 1.0)
  assert res.sampling_rate == 22050 and res.inference_duration_s > 0


@pytest.mark.parametrize("over", [dict(n_channels=96, n_layers=3, n_flows=4, n_early_every=2),
                                  dict(n_channels=384, n_layers=2, n_flows=2, n_early_every=1, n_early_size=2),
                                  dict(n_channels=30, n_layers=4, n_flows=4, n_early_every=2),
                                  dict(n_channels=64, n_layers=9, n_flows=2, n_early_every=1, n_early_size=2),
                                  dict(n_channels=128, n_layers=10, n_flows=2, n_early_every=1, n_early_size=2),
                                  dict(n_channels=64, n_mel_channels=24, n_layers=3, n_flows=4, n_early_every=2),
                                  dict(n_channels=96, n_mel_channels=72, n_layers=2, n_flows=2, n_early_every=1, n_early_size=2)],
                         ids=["c96", "c384", "c30", "l9", "l10", "m24", "c96m72"])
def test_hparam_envelope_channels_and_layers(over):
  """The reference takes any even n_channels, any n_layers and any n_mel_channels (model.py:75-113, :141-150,
  hparams.py:19-31).  Widths between the instantiated ones and mel counts that are not multiples of 16 run zero-padded
  (exact); 9 and 10 layers (dilations 256, 512) need 8 / 16 guard frames per utterance instead of 4.  infer and the no-grad forward against the CPU oracle, two utterances long enough for the largest
  dilation to reach real samples on both sides."""
  from oracle import torch_oracle as O
  hp = HParams(**over)
  sd = synthetic.make_state_dict(hp, seed=17)
  B, T, sigma = 2, 40, 0.8
  mel = synthetic.make_mel(B, T, seed=9)[:, :hp.n_mel_channels].contiguous()
  z_init, z_early = synthetic.make_noise(hp, B, 32 * T, seed=5)
  model = build_model(hp, sd)
  out = gpu_infer(model, mel, z_init, z_early, sigma)
  with torch.no_grad():
    ref = O.infer_ref(sd, mel, z_init, z_early, sigma, oracle_cfg_from_hp(hp))
  err = rms(out - ref)
  print(f"{over}: infer rms err {err:.3e} (signal rms {rms(ref):.3f})")
  assert err <= RMS_TOL
  g = torch.Generator().manual_seed(3)
  wav = torch.rand(B, 256 * T - 160, generator=g) * 0.6 - 0.3
  with torch.no_grad():
    z, log_s, log_det = model((mel.cuda(), wav.cuda()))
    z_ref, ls_ref, ld_ref = O.forward_ref(sd, mel, wav, oracle_cfg_from_hp(hp))
  assert rms(z.cpu() - z_ref) <= 2e-3 * max(1.0, rms(z_ref))
  for a, b in zip(log_s, ls_ref):
    assert rms(a.cpu() - b) <= 2e-3
