"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/waveglow_amd.h declares; host-only entry points (no GPU needed) behave."""
import ctypes as C
import os
import re

import pytest

from waveglow_amd import _lib, build

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def lib():
  build.build_library()
  return _lib.load()


def header_symbols():
  text = open(os.path.join(ROOT, "include", "waveglow_amd.h")).read()
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return sorted(set(re.findall(r"\b(wg_[a-z_]+)\s*\(", text)))


def test_header_symbols_all_exported_and_bound(lib):
  syms = header_symbols()
  assert len(syms) >= 14
  for s in syms:
    assert hasattr(lib, s), s
    assert s in _lib.SIGNATURES, f"{s} declared in the header but not bound in _lib.SIGNATURES"
  assert sorted(_lib.SIGNATURES) == syms


def test_create_validates_configuration(lib):
  h = C.c_void_p()
  bad = _lib.WgConfig(80, 12, 8, 4, 2, 8, 16, 3, 1024, 256)       # n_channels=16
  assert lib.wg_create(C.byref(bad), 0, C.byref(h)) == -1
  assert b"n_channels" in lib.wg_last_error()
  bad = _lib.WgConfig(80, 12, 4, 4, 2, 8, 256, 3, 1024, 256)      # n_group=4
  assert lib.wg_create(C.byref(bad), 0, C.byref(h)) == -1
  ok = _lib.WgConfig(80, 12, 8, 4, 2, 8, 256, 3, 1024, 256)
  assert lib.wg_create(C.byref(ok), 0, C.byref(h)) == 0
  # 2 upsample + 12 * (1 convinv + 6 + 8*4) tensors = 470 keys of the weight-norm-removed state_dict
  assert lib.wg_num_expected_tensors(h) == 470
  assert lib.wg_expected_tensor_name(h, 0) == b"upsample.weight"
  # algorithmic MACs per group-timestep, SURVEY.md section 8(d): 81 235 408 @ C=256
  assert int(lib.wg_macs_per_group_step(h)) == 81235408
  # workspace sizing is pure host arithmetic
  assert lib.wg_infer_workspace_bytes(h, 16, 864) > 16 * 27648 * (2 * 256 * 2 + 64)   # two x planes + Z + OUT
  assert lib.wg_infer_workspace_bytes(h, 0, 864) == 0
  # infer before finalize -> state error, not a crash
  import ctypes
  buf = ctypes.create_string_buffer(64)
  rc = lib.wg_infer(h, buf, buf, None, 0, 1.0, buf, 1, 1, 0, buf, 64, None)
  assert rc == -2
  # unknown tensor name rejected
  import numpy as np
  arr = np.zeros(4, dtype=np.float32)
  shp = (C.c_int64 * 1)(4)
  assert lib.wg_set_tensor(h, b"nope.weight", arr.ctypes.data, shp, 1) == -1
  assert lib.wg_destroy(h) == 0


def test_macs_c512(lib):
  h = C.c_void_p()
  cfg = _lib.WgConfig(80, 12, 8, 4, 2, 8, 512, 3, 1024, 256)
  assert lib.wg_create(C.byref(cfg), 0, C.byref(h)) == 0
  assert int(lib.wg_macs_per_group_step(h)) == 261355984
  lib.wg_destroy(h)


def test_train_entry_points_validate_arguments_without_a_gpu():
  """wg_train_* argument checks run before any device work: null members and bad geometry are reported, not crashed on."""
  import ctypes as C
  from waveglow_amd import _lib
  lib = _lib.load()
  cfg = _lib.WgConfig(80, 12, 8, 4, 2, 8, 256, 3, 1024, 256)
  h = C.c_void_p()
  assert lib.wg_create(C.byref(cfg), 0, C.byref(h)) == 0
  assert lib.wg_train_workspace_bytes(h, 32, 63, 16000) > 10 * 2 ** 30          # saved activations of configs[3]
  assert lib.wg_train_workspace_bytes(h, 32, 63, 16001) == 0                      # not a multiple of n_group
  assert lib.wg_train_workspace_bytes(h, 32, 10, 16000) == 0                      # upsampled mel shorter than audio
  w = _lib.WgTrainWeights()
  dummy = (C.c_char * 64)()
  ls = (C.c_void_p * 12)(*[C.addressof(dummy)] * 12)
  rc = lib.wg_train_forward(h, C.byref(w), C.addressof(dummy), C.addressof(dummy), C.addressof(dummy), ls, 1, 8, 2048, 0,
                            C.addressof(dummy), 1 << 40, None)
  assert rc == -1 and b"null member" in lib.wg_last_error()
  lib.wg_destroy(h)
