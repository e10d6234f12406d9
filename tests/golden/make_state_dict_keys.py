"""Dump the reference's state_dict key names + shapes (build container only; imports the reference).

  python tests/golden/make_state_dict_keys.py

Writes tests/golden/state_dict_keys.json: for the default HParams, the weight-normed form (what training
checkpoints hold) and the form after WaveGlow.remove_weightnorm (model.py:276-297).  Data only.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_import import import_reference  # noqa: E402

ref_model, ref_hparams, _ = import_reference()

out = {}
for tag, over in (("default", {}), ("c64_f4", dict(n_channels=64, n_flows=4, n_early_every=2, n_layers=3))):
  hp = ref_hparams.HParams(**over)
  m = ref_model.WaveGlow(hp)
  normed = {k: list(v.shape) for k, v in m.state_dict().items()}
  m = ref_model.WaveGlow.remove_weightnorm(m)
  dense = {k: list(v.shape) for k, v in m.state_dict().items()}
  out[tag] = {"hparams": over, "weight_normed": normed, "weight_norm_removed": dense,
              "n_params_weight_normed": sum(int(__import__("math").prod(s)) for s in normed.values())}
  print(tag, len(normed), len(dense), out[tag]["n_params_weight_normed"])
json.dump(out, open(os.path.join(HERE, "state_dict_keys.json"), "w"), indent=0, sort_keys=True)
