"""Fixtures for the wav helpers of the synthesize path, written by the reference's own ``audio_utils.py``
(build container only):  normalize_wav (audio_utils.py:67-95), convert_wav / float_to_wav (:26-32, :53-64),
is_overamp (:132-138).

  python tests/golden/make_golden_audio.py

Inputs are small arrays chosen to hit every branch (float32 / float64 / int16 / int32; silence; already normalised;
int minimum present; over-amplified floats).  The fixture stores inputs and the reference's outputs; float_to_wav's
output is the BYTES of the wav file it writes.
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from _ref_import import import_reference  # noqa: E402

import_reference()
import waveglow.audio_utils as ref_audio  # noqa: E402

rng = np.random.default_rng(12)
cases = {
  "f32_quiet": (rng.standard_normal(4001) * 0.11).astype(np.float32),
  "f32_loud": np.clip(rng.standard_normal(3000) * 0.5, -0.999, 0.999).astype(np.float32),
  "f32_unit": np.concatenate([(rng.standard_normal(500) * 0.2).astype(np.float32), np.float32([1.0, -0.25])]),
  "f32_silence": np.zeros(257, dtype=np.float32),
  "f32_overamp": (rng.standard_normal(1000) * 0.9 + np.float32(0.4)).astype(np.float32),
  "f64_denoised": (rng.standard_normal(2048) * 0.23),
  "i16_mid": (rng.standard_normal(3000) * 6000).astype(np.int16),
  "i16_has_min": np.concatenate([(rng.standard_normal(100) * 900).astype(np.int16), np.int16([-32768, 12])]),
  "i32_mid": (rng.standard_normal(1500) * 3.0e8).astype(np.int32),
}
out = {}
for name, x in cases.items():
  out[f"{name}/in"] = x
  out[f"{name}/is_overamp"] = np.array(bool(ref_audio.is_overamp(x)))
  over = bool(ref_audio.is_overamp(x))
  try:                                           # normalize_wav asserts on its own output (audio_utils.py:92-93)
    out[f"{name}/normalized"] = ref_audio.normalize_wav(x.copy())
    out[f"{name}/normalize_asserts"] = np.array(False)
  except AssertionError:
    out[f"{name}/normalize_asserts"] = np.array(True)
  if x.dtype in (np.float32, np.float64):
    out[f"{name}/as_int16"] = ref_audio.convert_wav(x.copy(), np.int16)
    with tempfile.TemporaryDirectory() as d:
      p = os.path.join(d, "a.wav")
      src = out.get(f"{name}/normalized", x)      # the CLI writes the normalised signal (inference_v2.py:124-130)
      ref_audio.float_to_wav(src, p)
      out[f"{name}/wav_bytes"] = np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
  else:
    out[f"{name}/as_float32"] = ref_audio.convert_wav(x.copy(), np.float32)
np.savez_compressed(os.path.join(HERE, "audio_utils.npz"), **out)
for k in sorted(out):
  print(k, out[k].dtype, out[k].shape)
