"""Loading the committed golden fixtures (tests/golden/*.npz) and regenerating the seeded
weights they were produced with.  Never touches /root/reference."""
import json
import os

import numpy as np
import torch

import golden_lib as gl

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLDEN, "manifest.json")) as _f:
    MANIFEST = json.load(_f)

SMALL = [n for n in MANIFEST["models"] if n.startswith("small_")]
CANON = ["qgpo_noise", "swiss_noise", "icfm_cond", "gvp_wide", "hopper_noise", "wideobs_noise"]


def model_info(name):
    kind, cfg = gl.model_zoo()[name]
    info = MANIFEST["models"][name]
    return kind, cfg, info


def weights(name, as_torch=True):
    info = MANIFEST["models"][name]
    w = gl.load_weights(name, [(k, tuple(s)) for k, s in info["key_shapes"]], GOLDEN)
    if as_torch:
        return {k: torch.from_numpy(v) for k, v in w.items()}
    return w


def load(fname):
    return dict(np.load(os.path.join(GOLDEN, fname)))


def T(a):
    return None if a is None else torch.from_numpy(np.asarray(a))


def cond_dim(cfg):
    return cfg["model"]["args"]["backbone"]["args"].get("condition_dim", None)
