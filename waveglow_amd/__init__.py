"""MI355X-native WaveGlow hot path (infer / forward) behind the reference's Python surface.

Public names mirror ``waveglow/__init__.py`` of stefantaubert/waveglow where the hot path touches them.
"""
from .hparams import HParams  # noqa: F401
from .model import WaveGlow  # noqa: F401
from .checkpoint import CheckpointWaveglow  # noqa: F401
from .synthesizer import InferenceResult, Synthesizer  # noqa: F401
from .model import WaveGlowLoss  # noqa: F401
from .audio import float_to_wav, normalize_wav  # noqa: F401


def __getattr__(name):
  # scipy-dependent / heavier pieces are imported on first use
  if name in ("train", "load_dataset", "Entry"):
    from . import training
    return getattr(training, name)
  if name in ("TacotronSTFT", "TSTFTHParams"):
    from . import taco_stft
    return getattr(taco_stft, name)
  if name == "Denoiser":
    from .denoiser import Denoiser
    return Denoiser
  raise AttributeError(name)
