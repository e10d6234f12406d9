"""MI355X-native WaveGlow hot path (infer / forward) behind the reference's Python surface.

Public names mirror ``waveglow/__init__.py`` of stefantaubert/waveglow where the hot path touches them.
"""
from .hparams import HParams  # noqa: F401
from .model import WaveGlow  # noqa: F401
from .checkpoint import CheckpointWaveglow  # noqa: F401
from .synthesizer import InferenceResult, Synthesizer  # noqa: F401
