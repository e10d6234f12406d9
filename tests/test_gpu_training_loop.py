"""GPU: mel front-end against the numpy oracle, and the training loop end to end -- the analogue of the reference's
own component test (src/waveglow_tests/test_training.py: random wav folders -> train()), plus resume."""
import wave
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_mel_frontend_matches_numpy_oracle():
  """TacotronSTFT.mel_spectrogram (taco_stft.py:84-104) through wg_stft_mel vs oracle/stft_oracle.py (fp64).
  Tolerance 2e-4 on the log-mel (fp32 MFMA chain of 1024 terms, then an 80 x 513 projection)."""
  from oracle import stft_oracle as S
  from waveglow_amd.hparams import HParams
  from waveglow_amd.taco_stft import TacotronSTFT
  hp = HParams()
  st = TacotronSTFT(hp, torch.device("cuda:0"))
  rng = np.random.default_rng(3)
  for B, N in ((3, 16000), (1, 5000), (2, 1024)):
    x = (rng.uniform(-0.8, 0.8, size=(B, N)) * np.linspace(0.2, 1.0, N)[None, :]).astype(np.float32)
    mel = st.mel_spectrogram(torch.from_numpy(x)).cpu().numpy()
    ref = S.mel_spectrogram(x, st.mel_basis.cpu().numpy())
    assert mel.shape == ref.shape == (B, 80, N // 256 + 1)
    err = np.abs(mel - ref).max()
    print(f"mel B={B} N={N}: max abs err {err:.2e}")
    assert err <= 2e-4
  one = st.get_mel_tensor(torch.from_numpy(x[0]))
  assert torch.equal(one.cpu(), torch.from_numpy(mel[0]))


def _random_wavs(folder: Path, n: int, seconds: float, seed: int):
  folder.mkdir(parents=True, exist_ok=True)
  rng = np.random.default_rng(seed)
  for i in range(n):
    data = np.int16(rng.uniform(-1.0, 1.0, size=int(seconds * 22050)) * 32767)
    with wave.open(str(folder / f"random_audio_{i + 1}.wav"), "w") as f:
      f.setnchannels(1)
      f.setsampwidth(2)
      f.setframerate(22050)
      f.writeframes(data.tobytes())


def test_train_component_and_resume(tmp_path):
  from waveglow_amd.checkpoint import CheckpointWaveglow
  from waveglow_amd.training import get_all_checkpoint_iterations, get_last_checkpoint, load_dataset, train
  trn, val, ckp = tmp_path / "trn", tmp_path / "val", tmp_path / "checkpoints"
  _random_wavs(trn, 6, 0.6, 1)
  _random_wavs(val, 2, 0.6, 2)
  custom = {"n_channels": "64", "n_layers": "3", "n_flows": "4", "n_early_every": "2", "batch_size": "2",
            "segment_length": "4096", "epochs": "2", "iters_per_checkpoint": "2", "learning_rate": "0.001"}
  dev = torch.device("cuda:0")
  losses = train(custom, tmp_path / "logs", load_dataset(trn), load_dataset(val), ckp, None, None, dev)
  assert len(losses) == 6 and all(np.isfinite(losses))          # 2 epochs x 3 batches
  assert get_all_checkpoint_iterations(ckp) == [1, 2, 3, 4, 6]  # first, every 2nd, epoch ends, last
  path, it = get_last_checkpoint(ckp)
  ck = CheckpointWaveglow.load(path, dev)
  assert it == 6 and ck.iteration == 6 and ck.get_hparams().n_channels == 64
  assert len(ck.state_dict) == 2 + 4 * (1 + 3 + 3 + 2 + 3 * 6)   # weight-normed set: conv 1, start 3, cond 3, end 2, 6 per layer
  # resume: one more epoch continues at iteration 7 with the optimiser state restored
  more = train({"epochs": "3"}, tmp_path / "logs", load_dataset(trn), load_dataset(val), ckp, ck, None, dev)
  assert len(more) == 3
  assert get_all_checkpoint_iterations(ckp) == [1, 2, 3, 4, 6, 8, 9]
  assert CheckpointWaveglow.load(get_last_checkpoint(ckp)[0], dev).iteration == 9
  # white noise cannot be modelled, but the likelihood still improves from the random start within a few steps
  assert np.mean(more) < losses[0]
