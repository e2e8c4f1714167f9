"""Builds this package's model classes for the golden zoo."""
import torch

import golden_lib as gl
import golden_util as gu
import generativerl_preview_amd as G

CLS = {"diffusion": G.DiffusionModel, "energy": G.EnergyConditionalDiffusionModel,
       "icfm": G.IndependentConditionalFlowModel}


def build(name, device="cpu"):
    kind, cfg = gl.model_zoo()[name]
    cfg = dict(cfg, device=device)
    m = CLS[kind](cfg)
    m.load_state_dict(gu.weights(name))
    return kind, cfg, m.to(device)


def fix_noise(model, x_T):
    """Replace the model's Gaussian draw with a fixed x_T (parity needs identical start states)."""
    def gen(batch_size=None):
        n = int(batch_size) if batch_size is not None else 1
        assert x_T.shape[0] == n, (x_T.shape, n)
        return x_T
    model.gaussian_generator = gen
