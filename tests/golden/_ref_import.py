"""Import the reference's own model code in the BUILD CONTAINER only (fixture generation).

/root/reference does not exist on the GPU box; nothing under tests/ imports this module
at test time.  The reference's package __init__ pulls in optional dependencies that are
absent here and are off the hot path, so empty stand-in modules
are registered for exactly those names before the import.
"""
import os
import sys
import types

REF_SRC = "/root/reference/src"
_STUBS = ["fastdtw", "fastdtw.fastdtw", "librosa", "librosa.util", "librosa.filters",
          "skimage", "skimage.metrics", "imageio", "mel_cepstral_distance", "wget",
          "gdown", "ordered_set"]


def import_reference():
  os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
  sys.dont_write_bytecode = True
  for name in _STUBS:
    if name not in sys.modules:
      m = types.ModuleType(name)
      m.__path__ = []  # allow submodule imports
      sys.modules[name] = m
  # names the reference imports *from* those modules at import time
  sys.modules["fastdtw.fastdtw"].fastdtw = lambda *a, **k: None
  sys.modules["fastdtw"].fastdtw = sys.modules["fastdtw.fastdtw"]
  for attr in ("pad_center", "tiny", "normalize"):
    setattr(sys.modules["librosa.util"], attr, lambda *a, **k: None)
  sys.modules["librosa.filters"].mel = lambda *a, **k: None
  sys.modules["librosa"].util = sys.modules["librosa.util"]
  sys.modules["librosa"].filters = sys.modules["librosa.filters"]
  sys.modules["skimage.metrics"].structural_similarity = lambda *a, **k: None
  sys.modules["ordered_set"].OrderedSet = set
  for attr in ("get_metrics_mels", "get_metrics_wavs", "compare_mel_spectrograms", "get_mcd_between_mel_spectograms"):
    setattr(sys.modules["mel_cepstral_distance"], attr, lambda *a, **k: None)
  if REF_SRC not in sys.path:
    sys.path.insert(0, REF_SRC)
  import waveglow.model as ref_model      # noqa: E402
  import waveglow.hparams as ref_hparams  # noqa: E402
  import waveglow.train as ref_train      # noqa: E402
  return ref_model, ref_hparams, ref_train
