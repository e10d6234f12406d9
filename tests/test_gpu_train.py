"""GPU parity of the TRAINING direction: forward with saved activations + the library's backward pass, against

* tests/golden/c64_grads.npz -- gradients of the reference's own ``loss.backward()`` (make_golden_grads.py), and
* oracle.grads_ref -- the CPU fp32 restatement (pinned bit-exact to that fixture in test_oracle_golden.py).

Tolerance: the kernels use fp16 MFMA operands, fp16 saved activations and fp16 gradient planes (loss-scaled) with
fp32 accumulation; per parameter tensor the gradient must match the fp32 reference to
``||g - g_ref|| <= GRAD_TOL * ||g_ref||`` (+ a small absolute floor for tensors whose gradient is ~0).
"""
import os

import numpy as np
import pytest
import torch

from _cases import oracle_cfg_from_hp
from waveglow_amd import synthetic
from waveglow_amd.hparams import HParams
from waveglow_amd.model import WaveGlow, WaveGlowLoss

pytestmark = pytest.mark.gpu

GRAD_TOL = 5e-3     # measured worst case 1.1e-3
FWD_TOL = 2e-3
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(autouse=True)
def _poisoned_gradient_buffer(monkeypatch):
  """Every test of this module starts its flat gradient buffer (waveglow_amd.train.GradBuffers, torch.empty in
  production) from NaN: an entry the library never writes -- also a PADDED one that pack_weights' backward slices off
  before any parity check sees it -- makes ``model.grad_finite`` (isfinite of the sum over the WHOLE buffer) false."""
  monkeypatch.setenv("WG_TRAIN_POISON_GRADS", "1")


def _setup(over, B, T, wseed, crop=96):
  hp = HParams(**over)
  sd = synthetic.to_weightnorm_form(synthetic.make_state_dict(hp, seed=wseed))
  mel = synthetic.make_mel(B, T, seed=1234 + B + T)
  S = 256 * T - crop
  g = torch.Generator().manual_seed(99 + T)
  wav = torch.rand(B, S, generator=g) * 0.6 - 0.3
  return hp, sd, mel, wav


def _gpu_step(hp, sd, mel, wav, sigma=1.0):
  model = WaveGlow(hp)
  model.load_state_dict(sd)
  model = model.to("cuda:0").train()
  model.zero_grad()
  y = model((mel.cuda(), wav.cuda()))
  loss = WaveGlowLoss(sigma)(y, None)
  loss.backward()
  torch.cuda.synchronize()
  assert bool(model.grad_finite), "the library left an entry of the (NaN-poisoned) gradient buffer unwritten"
  grads = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters()}
  return float(loss.detach()), y, grads


def _check(grads, ref, what):
  worst = []
  for name, g_ref in ref.items():
    g = grads[name]
    assert g.shape == g_ref.shape, name
    assert torch.isfinite(g).all(), name
    err = float((g - g_ref).norm())
    den = float(g_ref.norm())
    worst.append((err / max(den, 1e-12), name, err, den))
  worst.sort(reverse=True)
  for rel, name, err, den in worst[:8]:
    print(f"{what}: {name}: rel {rel:.3e} (err {err:.3e}, ref norm {den:.3e})")
  for rel, name, err, den in worst:
    assert err <= GRAD_TOL * den + 1e-7, f"{name}: gradient error {err:.3e} vs norm {den:.3e}"


def test_train_step_matches_reference_fixture_and_oracle():
  from oracle import torch_oracle as O
  over = dict(n_channels=64, n_layers=4, n_flows=6, n_early_every=2)
  hp, sd, mel, wav = _setup(over, 2, 12, 5)
  loss, y, grads = _gpu_step(hp, sd, mel, wav)
  fx = np.load(os.path.join(HERE, "golden", "c64_grads.npz"))
  print(f"loss gpu {loss:.6f} reference {float(fx['loss']):.6f}")
  assert abs(loss - float(fx["loss"])) <= 2e-3 * max(1.0, abs(float(fx["loss"])))
  # forward outputs vs the oracle
  cfg = oracle_cfg_from_hp(hp)
  loss_ref, g_ref = O.grads_ref(sd, mel, wav, cfg, 1.0)
  assert abs(float(loss_ref) - float(fx["loss"])) < 1e-6
  _check(grads, g_ref, "c64")
  # and directly against what the reference wrote down
  for key in fx.files:
    if key.startswith("full/"):
      name = key[5:]
      ref = torch.from_numpy(fx[key])
      err = float((grads[name] - ref).norm())
      assert err <= GRAD_TOL * float(ref.norm()) + 1e-7, name
    elif key.startswith("norm/"):
      name = key[5:]
      assert abs(float(grads[name].norm()) - float(fx[key])) <= GRAD_TOL * float(fx[key]) + 1e-7, name


def test_train_forward_outputs_match_oracle():
  from oracle import torch_oracle as O
  over = dict(n_channels=64, n_layers=4, n_flows=6, n_early_every=2)
  hp, sd, mel, wav = _setup(over, 2, 12, 5)
  model = WaveGlow(hp)
  model.load_state_dict(sd)
  model = model.to("cuda:0").train()
  z, log_s, log_det = model((mel.cuda(), wav.cuda()))
  dense = {k: v.detach().cpu() for k, v in model.dense_state().items()}
  z_ref, ls_ref, ld_ref = O.forward_ref(dense, mel, wav, oracle_cfg_from_hp(hp))
  ez = float((z.cpu() - z_ref).pow(2).mean().sqrt())
  print("z rms err", ez)
  assert ez <= FWD_TOL
  for a, b in zip(log_s, ls_ref):
    assert float((a.cpu() - b).pow(2).mean().sqrt()) <= FWD_TOL
  for a, b in zip(log_det, ld_ref):
    assert abs(float(a) - float(b)) <= 1e-3 * max(1.0, abs(float(b)))


def test_train_step_c256_two_utterances():
  """Full-width model (256 channels, 8 layers, 12 flows), ragged length: every parameter gradient vs the oracle."""
  from oracle import torch_oracle as O
  hp, sd, mel, wav = _setup(dict(), 2, 9, 3, crop=40)
  loss, y, grads = _gpu_step(hp, sd, mel, wav)
  loss_ref, g_ref = O.grads_ref(sd, mel, wav, oracle_cfg_from_hp(hp), 1.0)
  print(f"loss gpu {loss:.6f} oracle {float(loss_ref):.6f}")
  assert abs(loss - float(loss_ref)) <= 2e-3 * max(1.0, abs(float(loss_ref)))
  _check(grads, g_ref, "c256")


@pytest.mark.parametrize("force_bn", ["128", "64"])
def test_train_step_both_layer_tile_widths(force_bn, monkeypatch):
  """The fused layer forward and the two dgrad GEMMs run on the WN-layer kernel, whose tile width (128 or 64 columns)
  is chosen by workload size: small test shapes would only ever see 64.  Both widths, 256 channels (8 waves, the
  pipelined epilogue with saved activations) and 64 channels (2 waves), against the oracle."""
  from oracle import torch_oracle as O
  monkeypatch.setenv("WG_FORCE_BN", force_bn)
  for over, B, T, seed in ((dict(n_layers=3, n_flows=2, n_early_every=1, n_early_size=2), 3, 11, 4),
                           (dict(n_channels=64, n_layers=8, n_flows=4, n_early_every=2), 5, 9, 6)):
    hp, sd, mel, wav = _setup(over, B, T, seed, crop=56)
    loss, y, grads = _gpu_step(hp, sd, mel, wav)
    loss_ref, g_ref = O.grads_ref(sd, mel, wav, oracle_cfg_from_hp(hp), 1.0)
    assert abs(loss - float(loss_ref)) <= 2e-3 * max(1.0, abs(float(loss_ref)))
    _check(grads, g_ref, f"bn{force_bn}/c{hp.n_channels}")


@pytest.mark.parametrize("over,B", [(dict(n_layers=3, n_flows=3, n_early_every=1, n_early_size=2), 2),
                                    (dict(n_channels=64, n_layers=8, n_flows=4, n_early_every=2), 4)])
def test_train_step_half_batch_chains(over, B, monkeypatch):
  """Large batches run as two half-batch chains on two streams, with the weight-gradient launches on a third (train_api.cpp:
  setup, wg_train_backward_flows); small test shapes would never choose that, so it is forced here.  The same step with
  everything serialised on one stream must give bit-identical outputs and gradients (the streams only reorder launches
  that do not depend on each other -- any difference is a missing dependency), and both match the oracle."""
  from oracle import torch_oracle as O
  hp, sd, mel, wav = _setup(over, B, 11, 9, crop=56)
  monkeypatch.setenv("WG_TRAIN_HALVES", "2")
  monkeypatch.setenv("WG_TRAIN_BWD_HALVES", "2")
  monkeypatch.setenv("WG_TRAIN_SERIAL", "1")
  loss_s, y_s, g_s = _gpu_step(hp, sd, mel, wav)
  monkeypatch.setenv("WG_TRAIN_SERIAL", "0")
  for rep in range(3):
    monkeypatch.setenv("WG_TRAIN_BWD_HALVES", "2" if rep < 2 else "1")       # the default: one chain + the weight-gradient stream
    loss_c, y_c, g_c = _gpu_step(hp, sd, mel, wav)
    assert loss_c == loss_s
    assert torch.equal(y_c[0].detach().cpu(), y_s[0].detach().cpu())
    for name in g_s:
      assert torch.equal(g_c[name], g_s[name]), f"{name}: chains differ from the serial run (rep {rep})"
  loss_ref, g_ref = O.grads_ref(sd, mel, wav, oracle_cfg_from_hp(hp), 1.0)
  assert abs(loss_c - float(loss_ref)) <= 2e-3 * max(1.0, abs(float(loss_ref)))
  _check(g_c, g_ref, f"chains c{hp.n_channels}")


@pytest.mark.parametrize("slabs", ["1", "3,5", "7", "32", "50,64", "128"])
def test_train_step_every_slab_shape(slabs, monkeypatch):
  """The weight-gradient kernel cuts the rows (32 phases x Rp) into n_slabs contiguous ranges of 32-row steps, one fp32
  partial tile per workgroup and range; the launcher picks n_slabs = CUs / tiles.  Here (Rp = 128: 4 steps per phase, 128
  in all) pinned to: everything in one slab, ranges that start and end inside phases (3, 7, 50), one phase per slab (32)
  and one step per slab (128), the two jobs of a layer with equal or different slab counts -- against the oracle."""
  from oracle import torch_oracle as O
  monkeypatch.setenv("WG_TRAIN_SLABS", slabs)
  hp, sd, mel, wav = _setup(dict(n_layers=3, n_flows=2, n_early_every=1, n_early_size=2), 3, 11, 8, crop=56)
  loss, y, grads = _gpu_step(hp, sd, mel, wav)
  loss_ref, g_ref = O.grads_ref(sd, mel, wav, oracle_cfg_from_hp(hp), 1.0)
  assert abs(loss - float(loss_ref)) <= 2e-3 * max(1.0, abs(float(loss_ref)))
  _check(grads, g_ref, f"slabs {slabs}")


@pytest.mark.parametrize("ct,mt", [("2", "5"), ("3", "5"), ("2", "1"), ("3", "1"), ("4", "1"), ("6", "1")])
def test_train_step_every_tile_width(ct, mt, monkeypatch):
  """Rp = 384 rows per phase divides by 64, 96, 128 and 192: every tile shape of the plane GEMM (column width x all 640
  rows in one workgroup or 128-row groups) gives the same gradients (the launcher normally picks the one with the fewest
  rounds)."""
  from oracle import torch_oracle as O
  monkeypatch.setenv("WG_TRAIN_CT", ct)
  monkeypatch.setenv("WG_TRAIN_MT", mt)
  over = dict(n_channels=64, n_layers=3, n_flows=4, n_early_every=2)
  hp, sd, mel, wav = _setup(over, 6, 50, 11, crop=72)
  loss, y, grads = _gpu_step(hp, sd, mel, wav)
  loss_ref, g_ref = O.grads_ref(sd, mel, wav, oracle_cfg_from_hp(hp), 1.0)
  assert abs(loss - float(loss_ref)) <= 2e-3 * max(1.0, abs(float(loss_ref)))
  _check(grads, g_ref, f"ct{ct} mt{mt}")


def test_two_forwards_before_backward_and_double_backward():
  """Gradient accumulation pattern: forward, forward, backward, backward -- the second forward gets its own training
  workspace, so both graphs back-propagate correctly (sum of the two gradients = 2x the single-step gradient for the
  same batch); a second backward through the same graph fails loudly (the activations are released)."""
  from waveglow_amd._lib import WgError
  over = dict(n_channels=64, n_layers=3, n_flows=4, n_early_every=2)
  hp, sd, mel, wav = _setup(over, 2, 6, 2)
  _, _, single = _gpu_step(hp, sd, mel, wav)
  model = WaveGlow(hp)
  model.load_state_dict(sd)
  model = model.to("cuda:0").train()
  crit = WaveGlowLoss(1.0)
  l1 = crit(model((mel.cuda(), wav.cuda())), None)
  l2 = crit(model((mel.cuda(), wav.cuda())), None)
  l1.backward()
  l2.backward()
  torch.cuda.synchronize()
  for name, p in model.named_parameters():
    ref = 2.0 * single[name]
    assert float((p.grad.cpu() - ref).norm()) <= 1e-5 * float(ref.norm()) + 1e-9, name
  with pytest.raises((WgError, RuntimeError)):
    l2.backward()
  assert len(model._engine._train_pool) == 2 and not any(e["busy"] for e in model._engine._train_pool)
  # a graph that is dropped without backward() (loss only logged) gives its workspace back as well
  for _ in range(3):
    l3 = crit(model((mel.cuda(), wav.cuda())), None)
    del l3
  assert len(model._engine._train_pool) == 2 and not any(e["busy"] for e in model._engine._train_pool)


def test_gradient_allreduce_over_rccl_single_rank():
  """The exchange step on the device: bucketed all-reduce through RCCL ("nccl") in a one-rank group leaves the
  gradients of a real training step unchanged (the two-rank averaging itself is covered on CPU with gloo)."""
  import socket
  import torch.distributed as dist
  from waveglow_amd.distributed import GradientAllReducer
  over = dict(n_channels=64, n_layers=3, n_flows=4, n_early_every=2)
  hp, sd, mel, wav = _setup(over, 2, 6, 2)
  model = WaveGlow(hp)
  model.load_state_dict(sd)
  model = model.to("cuda:0").train()
  WaveGlowLoss(1.0)(model((mel.cuda(), wav.cuda())), None).backward()
  before = [p.grad.clone() for p in model.parameters()]
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  port = s.getsockname()[1]
  s.close()
  dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                          device_id=torch.device("cuda:0"))
  try:
    red = GradientAllReducer(model.parameters(), bucket_bytes=1 << 20)
    assert len(red.buckets) > 3
    red.reduce(force=True)
    torch.cuda.synchronize()
  finally:
    dist.destroy_process_group()
  for a, p in zip(before, model.parameters()):
    assert torch.equal(a, p.grad)


@pytest.mark.parametrize("B", [8, 32])
def test_full_size_directional_derivative_and_forward_consistency(B):
  """BASELINE configs[3] shapes (256 channels, 16 000-sample segments; batch 8, and the config's real per-GPU batch of 32:
  the launcher then picks two forward chains and 22 GB of saved planes by itself), where no CPU
  oracle finishes in seconds: size-independent checks.  (1) the training forward (saved activations, unfolded cond
  path) and the inference-style forward (folded weights, wn_layer_kernel) agree on the loss; (2) the gradient of the
  library's backward predicts the central finite difference of that loss along the gradient direction."""
  hp, sd, mel, wav = _setup(dict(), B, 63, 7, crop=256 * 63 - 16000)
  assert wav.shape == (B, 16000)
  model = WaveGlow(hp)
  model.load_state_dict(sd)
  model = model.to("cuda:0").train()
  crit = WaveGlowLoss(1.0)
  mel_d, wav_d = mel.cuda(), wav.cuda()
  loss = crit(model((mel_d, wav_d)), None)
  loss.backward()
  assert bool(model.grad_finite)
  grads = [p.grad.detach().clone() for p in model.parameters()]
  gnorm = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads)))
  assert np.isfinite(gnorm) and gnorm > 0

  def loss_at(eps):
    with torch.no_grad():
      for p, g in zip(model.parameters(), grads):
        p.add_(g, alpha=eps / gnorm)
      val = float(crit(model((mel_d, wav_d)), None))      # no-grad path: wg_forward on re-derived (folded) weights
      for p, g in zip(model.parameters(), grads):
        p.sub_(g, alpha=eps / gnorm)
    return val

  l0 = loss_at(0.0)
  print(f"loss train-forward {float(loss.detach()):.6f}  inference-forward {l0:.6f}  |grad| {gnorm:.4e}")
  assert abs(l0 - float(loss.detach())) <= 2e-4 * max(1.0, abs(l0))
  eps = 0.02 / gnorm if gnorm > 1 else 0.02                 # a step that changes the loss by about 0.02 * |grad|
  fd = (loss_at(eps) - loss_at(-eps)) / (2 * eps)
  print(f"directional derivative: finite difference {fd:.5e}  vs  |grad| {gnorm:.5e}")
  assert abs(fd - gnorm) <= 0.03 * gnorm


@pytest.mark.parametrize("channels", [128, 512])
def test_train_step_other_widths(channels):
  """128 and 512 channels (1 and 4 row groups of the plane GEMM, 2 / 8 weight-gradient row tiles), small depth."""
  from oracle import torch_oracle as O
  over = dict(n_channels=channels, n_layers=2, n_flows=2, n_early_every=1, n_early_size=2)
  hp, sd, mel, wav = _setup(over, 2, 7, 13, crop=24)
  loss, y, grads = _gpu_step(hp, sd, mel, wav)
  loss_ref, g_ref = O.grads_ref(sd, mel, wav, oracle_cfg_from_hp(hp), 1.0)
  assert abs(loss - float(loss_ref)) <= 2e-3 * max(1.0, abs(float(loss_ref)))
  _check(grads, g_ref, f"c{channels}")


def test_backward_with_fused_per_flow_allreduce_single_rank():
  """Data-parallel mode of the autograd node: backward cut at flow boundaries (wg_train_backward_flows) with one RCCL
  all-reduce per flow queued behind it.  In a one-rank group the result must equal the single-call backward."""
  import socket
  import torch.distributed as dist
  from waveglow_amd.train import enable_data_parallel
  over = dict(n_channels=64, n_layers=3, n_flows=6, n_early_every=2)
  hp, sd, mel, wav = _setup(over, 2, 6, 2)
  _, _, ref = _gpu_step(hp, sd, mel, wav)
  model = WaveGlow(hp)
  model.load_state_dict(sd)
  model = model.to("cuda:0").train()
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  port = s.getsockname()[1]
  s.close()
  dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                          device_id=torch.device("cuda:0"))
  try:
    assert not enable_data_parallel(model)            # a single process keeps the plain path ...
    assert enable_data_parallel(model, force=True)    # ... unless forced
    WaveGlowLoss(1.0)(model((mel.cuda(), wav.cuda())), None).backward()
    torch.cuda.synchronize()
  finally:
    dist.destroy_process_group()
  assert bool(model.grad_finite)      # poisoned buffer (module fixture): the per-flow calls together write every entry
  for name, p in model.named_parameters():
    assert torch.equal(p.grad.cpu(), ref[name]), name


def test_train_step_small_mel_dimension():
  """n_mel_channels = 32 (M8 = 256: 2 row groups of the upsample GEMM, 4 conditioning chunks), one early flow."""
  from oracle import torch_oracle as O
  over = dict(n_channels=64, n_layers=2, n_flows=3, n_early_every=2, n_mel_channels=32)
  hp = HParams(**over)
  sd = synthetic.to_weightnorm_form(synthetic.make_state_dict(hp, seed=21))
  mel = synthetic.make_mel(3, 8, n_mel=32, seed=4)
  wav = torch.rand(3, 256 * 8 - 40, generator=torch.Generator().manual_seed(8)) * 0.6 - 0.3
  loss, y, grads = _gpu_step(hp, sd, mel, wav)
  loss_ref, g_ref = O.grads_ref(sd, mel, wav, oracle_cfg_from_hp(hp), 1.0)
  assert abs(loss - float(loss_ref)) <= 2e-3 * max(1.0, abs(float(loss_ref)))
  _check(grads, g_ref, "mel32")


def test_train_step_configs3_shapes_against_reference_summary():
  """BASELINE configs[3] shapes (256 channels, 16 000-sample segments, 63 mel frames; batch 2 as in the reference's CPU
  probe): loss and the norm / sum / first 8 values of all 686 parameter gradients written by the reference's own
  ``loss.backward()`` (tests/golden/make_golden_grads.py cfg4_b2)."""
  fx = np.load(os.path.join(HERE, "golden", "cfg4_b2_grads.npz"))
  hp, sd, mel, wav = _setup(dict(), 2, 63, 0, crop=256 * 63 - 16000)
  loss, y, grads = _gpu_step(hp, sd, mel, wav)
  print(f"loss gpu {loss:.6f} reference {float(fx['loss']):.6f}")
  assert abs(loss - float(fx["loss"])) <= 2e-3 * max(1.0, abs(float(fx["loss"])))
  worst = 0.0
  n = 0
  for key in fx.files:
    if not key.startswith("norm/"):
      continue
    name = key[5:]
    g = grads[name]
    ref_norm = float(fx[key])
    assert abs(float(g.norm()) - ref_norm) <= GRAD_TOL * ref_norm + 1e-7, name
    head = torch.from_numpy(fx["head/" + name])
    err = float((g.flatten()[:head.numel()] - head).norm())
    assert err <= GRAD_TOL * max(float(head.norm()), 1e-3 * ref_norm) + 1e-7, name
    worst = max(worst, abs(float(g.norm()) - ref_norm) / max(ref_norm, 1e-12))
    n += 1
  assert n == 686
  print(f"686 gradients: worst norm deviation {worst:.2e}")


def test_custom_grad_scale_and_finite_check(monkeypatch):
  """model.grad_scale only moves the fp16 gradient planes' exponent: results agree with the automatic scale; an absurd
  scale overflows the planes and WG_TRAIN_CHECK_FINITE=1 reports it instead of handing nan gradients to the optimiser."""
  from waveglow_amd._lib import WgError
  over = dict(n_channels=64, n_layers=3, n_flows=4, n_early_every=2)
  hp, sd, mel, wav = _setup(over, 2, 6, 2)
  _, _, ref = _gpu_step(hp, sd, mel, wav)
  model = WaveGlow(hp)
  model.load_state_dict(sd)
  model = model.to("cuda:0").train()
  model.grad_scale = 256.0
  WaveGlowLoss(1.0)(model((mel.cuda(), wav.cuda())), None).backward()
  for name, p in model.named_parameters():
    assert float((p.grad.cpu() - ref[name]).norm()) <= 5e-3 * float(ref[name].norm()) + 1e-8, name
  assert bool(model.grad_finite)                    # device-side flag over ALL gradient tensors, read by train()
  model.zero_grad()
  model.grad_scale = 1e30
  WaveGlowLoss(1.0)(model((mel.cuda(), wav.cuda())), None).backward()
  assert not bool(model.grad_finite)
  # train() hands the flag to the fused optimiser (found_inf): the update is skipped on the device, before the host looks
  from waveglow_amd.training import load_optimizer
  opt = load_optimizer(model.parameters(), hp, None)
  assert any(g.get("fused") for g in opt.param_groups)
  before = [p.detach().clone() for p in model.parameters()]
  opt.found_inf = (~model.grad_finite).to(torch.float32).reshape(())
  opt.step()
  assert all(torch.equal(a, p.detach()) for a, p in zip(before, model.parameters()))
  monkeypatch.setenv("WG_TRAIN_CHECK_FINITE", "1")
  model.zero_grad()
  with pytest.raises(WgError):
    WaveGlowLoss(1.0)(model((mel.cuda(), wav.cuda())), None).backward()


@pytest.mark.parametrize("channels,wn", [(64, False), (256, False), (64, True), (256, True)])
def test_prepare_matches_torch_packing(channels, wn):
  """wg_train_prepare (the library reads the module's own parameter tensors: weight norm, W_end x W_skip fold, permutations,
  gate pre-scale, fragment orders -- train_prep.hip, pack_kernel) against the same computation written as torch ops
  (waveglow_amd/train.py: pack_weights, wn_forward_fragments, plain_fragments, to_fragments -- pinned element by element to
  the header's index maps in tests/test_host_cpu.py).  Dense weights (after remove_weightnorm): every tensor that is pure
  data movement is bit-identical; with weight norm (and for the fold) the fp32 summation order differs from torch's, so
  fp16 values may differ in the last place."""
  from waveglow_amd import train as T
  hp = HParams(n_channels=channels, n_layers=3, n_flows=2, n_early_every=1, n_early_size=2)
  sd = synthetic.to_weightnorm_form(synthetic.make_state_dict(hp, seed=21))
  model = WaveGlow(hp)
  model.load_state_dict(sd)
  if not wn:
    model = WaveGlow.remove_weightnorm(model)
  model = model.to("cuda:0").train()
  eng = model._get_engine(torch.device("cuda:0"), need_weights=False)
  NW = int(eng.lib.wg_wn_waves(channels))
  with torch.no_grad():
    packed = [t.detach() for t in T.pack_weights(model)]
    names, tensors, is_wn = T.canonical_params(model, eng)
    assert is_wn == wn and len(tensors) == len(list(model.parameters()))
    w = T._Weights(model, tensors, wn, model.flow_channels(), eng, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    Cc, M8 = channels, hp.n_mel_channels * 8
    pm = T._perms(Cc, M8, packed[0].device)
    a1, a1c, b1s, a2, es = T.wn_forward_fragments(packed[0], packed[1], packed[2], packed[4], pm, NW)
    FL = packed[0].shape[0]
    w1h = packed[0].half()
    w1c = w1h[:, :, 3 * Cc:].index_select(1, pm.c2).index_select(2, pm.m8)
    wat_m = torch.cat([packed[2].transpose(1, 2).index_select(2, pm.c),
                       torch.nn.functional.pad(packed[4].half().float(), (0, 0, 0, 56)).transpose(1, 2)], 2)
    want = {"a1": a1, "a1c": a1c, "a2": a2, "es": es,
            "wat": T.plain_fragments(wat_m.half(), NW),
            "wbt": T.plain_fragments(torch.cat([w1h[:, :, t * Cc:(t + 1) * Cc].transpose(1, 2).index_select(2, pm.c2)
                                                for t in range(3)], 2), NW),
            "wct": T.to_fragments(w1c.permute(2, 0, 1).reshape(-1, FL * 2 * Cc), pm.c2p),
            "wup": T.to_fragments(packed[5].index_select(1, pm.m8).half(), pm.c2p)}
    for name, t in want.items():
      got = getattr(w, name)
      assert got.numel() == t.numel(), name
      if not wn and name not in ("es", "wat"):
        assert torch.equal(got.view(-1).view(torch.int16), t.reshape(-1).view(torch.int16)), name
      else:
        g32, t32 = got.view(-1).float(), t.reshape(-1).float()
        if name == "es":      # [FL, s, l4, 16 rows, 8]: rows 0-7 the hi fp16 half, 8-15 the lo half -- their SUM is the value
          g32, t32 = [x.view(FL, Cc // 32, 4, 2, 8, 8).sum(3).reshape(-1) for x in (g32, t32)]
        tol = 1.5e-3 * t32.abs() + 1e-6
        assert bool(((g32 - t32).abs() <= tol).all()), (name, float((g32 - t32).abs().max()))
    assert torch.allclose(w.b1, b1s, rtol=1e-6, atol=1e-7)
    assert torch.allclose(w.b2, packed[3], rtol=1e-6, atol=1e-7)
    assert torch.equal(w.bup, packed[6].index_select(0, pm.m8))
    start5, out_init, w1x1 = packed[7].index_select(2, pm.c), packed[8], packed[9]
    for k, c in enumerate(model.flow_channels()):
      assert torch.allclose(w.wstart[k].view(Cc, c // 2), start5[k, :c // 2].transpose(0, 1), rtol=2e-6, atol=1e-7), k
      assert torch.equal(w.bstart[k], start5[k, 4]), k
      assert torch.allclose(w.out_init[k], out_init[k], rtol=1e-5, atol=1e-6), k
      assert torch.equal(w.w1x1[k].view(c, c), w1x1[k, :c, :c]), k


def test_training_on_dense_weights_after_remove_weightnorm():
  """The training direction also takes a model whose weight norm has been removed (plain ``weight`` parameters): the
  gradients are then the dense weights' own (oracle: the same state dict in dense form)."""
  from oracle import torch_oracle as O
  over = dict(n_channels=64, n_layers=3, n_flows=4, n_early_every=2)
  hp, sd, mel, wav = _setup(over, 2, 6, 2)
  model = WaveGlow.remove_weightnorm(WaveGlow(hp))
  dense = synthetic.make_state_dict(hp, seed=2)
  model.load_state_dict(dense)
  model = model.to("cuda:0").train()
  loss = WaveGlowLoss(1.0)(model((mel.cuda(), wav.cuda())), None)
  loss.backward()
  torch.cuda.synchronize()
  assert bool(model.grad_finite)
  loss_ref, g_ref = O.grads_ref(dense, mel, wav, oracle_cfg_from_hp(hp), 1.0)
  assert abs(float(loss.detach()) - float(loss_ref)) <= 2e-3 * max(1.0, abs(float(loss_ref)))
  _check({n: p.grad.detach().float().cpu() for n, p in model.named_parameters()}, g_ref, "dense")


@pytest.mark.parametrize("channels", [64, 256])
def test_unfused_dgrad_launches_give_the_same_gradients(channels, monkeypatch):
  """Default: the d x launch of layer i carries d acts + the gate derivative of layer i - 1 behind it (wn_layer_kernel
  MODE 4: the d x tile goes through LDS into the next GEMM).  WG_TRAIN_NO_FUSE=1 runs them as launches of their own (MODE 2
  + MODE 3): same fp16 inputs to every GEMM except that the fused path feeds the d x tile to the second GEMM before it is
  rounded through memory -- identically rounded, so the gradients agree to the last bit."""
  over = dict(n_channels=channels, n_layers=3, n_flows=2, n_early_every=1, n_early_size=2)
  hp, sd, mel, wav = _setup(over, 2, 9, 4, crop=40)
  _, _, fused = _gpu_step(hp, sd, mel, wav)
  monkeypatch.setenv("WG_TRAIN_NO_FUSE", "1")
  _, _, plain = _gpu_step(hp, sd, mel, wav)
  for name in fused:
    assert torch.equal(fused[name], plain[name]), name
