"""ctypes binding of libwaveglow_amd.so (C ABI: include/waveglow_amd.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``waveglow_amd.build.build_library()``.
There is no fallback: if it is missing or fails to load, every compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WAVEGLOW_AMD_LIB", os.path.join(_HERE, "csrc", "libwaveglow_amd.so"))

WG_F32, WG_F16 = 0, 1


class WgConfig(C.Structure):
  _fields_ = [(n, C.c_int32) for n in (
    "n_mel_channels", "n_flows", "n_group", "n_early_every", "n_early_size", "n_layers",
    "n_channels", "kernel_size", "upsample_kernel", "upsample_stride")]


class WgTrainWeights(C.Structure):
  """wg_train_weights (include/waveglow_amd.h): device pointers; the last four are arrays of n_flows pointers."""
  _fields_ = [(n, C.c_void_p) for n in (
    "a1", "a1c", "b1", "a2", "b2", "es", "wat", "wbt", "wct", "wup", "bup", "wstart", "bstart", "out_init", "w1x1")]


class WgTrainPlain(C.Structure):
  """wg_train_plain: the natural-order fp32 matrices wg_train_pack reads."""
  _fields_ = [(n, C.c_void_p) for n in ("w1", "w2", "wes", "wup")]


class WgTrainGrads(C.Structure):
  """wg_train_grads: device pointers; the last three are arrays of n_flows pointers."""
  _fields_ = [(n, C.c_void_p) for n in (
    "dw1", "db1", "dw2", "db2", "dwes", "dwup", "dbup", "dstart", "dout_init", "dw1x1")] + [
    ("layer_stride", C.c_int64), ("flow_stride", C.c_int64)]


class WgError(RuntimeError):
  pass


_lib: Optional[C.CDLL] = None

# name -> (restype, argtypes); every symbol declared in include/waveglow_amd.h
SIGNATURES = {
  "wg_version": (C.c_char_p, []),
  "wg_last_error": (C.c_char_p, []),
  "wg_create": (C.c_int, [C.POINTER(WgConfig), C.c_int, C.POINTER(C.c_void_p)]),
  "wg_destroy": (C.c_int, [C.c_void_p]),
  "wg_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32]),
  "wg_num_expected_tensors": (C.c_int, [C.c_void_p]),
  "wg_expected_tensor_name": (C.c_char_p, [C.c_void_p, C.c_int32]),
  "wg_finalize": (C.c_int, [C.c_void_p]),
  "wg_infer_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32]),
  "wg_forward_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
  "wg_infer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_float,
                         C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
  "wg_infer_ragged": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int32,
                                C.c_float, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t,
                                C.c_void_p]),
  "wg_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p),
                           C.POINTER(C.c_float), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                           C.c_void_p, C.c_size_t, C.c_void_p]),
  "wg_loss": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int32,
                        C.POINTER(C.c_float), C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
  "wg_loss_dev": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int32,
                            C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
  "wg_macs_per_group_step": (C.c_double, [C.c_void_p]),
  "wg_train_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
  "wg_wn_waves": (C.c_int32, [C.c_int32]),
  "wg_train_pack": (C.c_int, [C.c_void_p, C.POINTER(WgTrainPlain), C.POINTER(WgTrainWeights), C.c_void_p]),
  "wg_train_param_count": (C.c_int32, [C.c_void_p, C.c_int32]),
  "wg_train_param_name": (C.c_char_p, [C.c_void_p, C.c_int32, C.c_int32]),
  "wg_train_param_numel": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32]),
  "wg_train_prepare_bytes": (C.c_size_t, [C.c_void_p]),
  "wg_train_prepare": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.POINTER(WgTrainWeights), C.c_void_p,
                                 C.c_size_t, C.c_void_p]),
  "wg_train_param_grads": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.POINTER(WgTrainGrads), C.c_void_p,
                                     C.c_size_t, C.c_void_p, C.c_void_p]),
  "wg_train_forward": (C.c_int, [C.c_void_p, C.POINTER(WgTrainWeights), C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                 C.c_size_t, C.c_void_p]),
  "wg_train_backward": (C.c_int, [C.c_void_p, C.POINTER(WgTrainWeights), C.POINTER(WgTrainGrads), C.c_void_p,
                                  C.POINTER(C.c_void_p), C.c_float, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_void_p, C.c_size_t, C.c_void_p]),
  "wg_train_backward_flows": (C.c_int, [C.c_void_p, C.POINTER(WgTrainWeights), C.POINTER(WgTrainGrads), C.c_void_p,
                                        C.POINTER(C.c_void_p), C.c_float, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_void_p]),
  "wg_stft_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
  "wg_stft_destroy": (C.c_int, [C.c_void_p]),
  "wg_stft_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32]),
  "wg_stft_denoise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int32,
                                C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
  "wg_stft_mel_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32]),
  "wg_stft_mel": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                            C.c_void_p, C.c_size_t, C.c_void_p]),
  "wg_debug_set_stamp_buffer": (C.c_int, [C.c_void_p, C.c_void_p]),
  "wg_profile_enable": (C.c_int, [C.c_void_p, C.c_int32]),
  "wg_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int32]),
}


def load() -> C.CDLL:
  """Load the HIP library or raise -- never falls back to another implementation."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.isfile(LIB_PATH):
    raise WgError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                  "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
  lib = C.CDLL(LIB_PATH)
  for name, (res, args) in SIGNATURES.items():
    fn = getattr(lib, name)
    fn.restype = res
    fn.argtypes = args
  _lib = lib
  return lib


def device_index(device) -> int:
  """Ordinal of a 'cuda' torch.device; an index-less ``torch.device('cuda')`` means torch's CURRENT device (under
  LOCAL_RANK > 0 that is not GPU 0)."""
  import torch
  device = torch.device(device)
  return device.index if device.index is not None else torch.cuda.current_device()


def check(rc: int) -> None:
  if rc != 0:
    raise WgError(f"libwaveglow_amd error {rc}: {load().wg_last_error().decode()}")
