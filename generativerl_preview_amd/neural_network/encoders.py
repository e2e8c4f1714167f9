"""Encoder registry and the Gaussian-Fourier time encoder (grl/neural_network/encoders.py:8-86,299-306).

The modules here own the parameters under the reference's state_dict names; the arithmetic of the
hot path runs in the HIP library (the Fourier embedding is evaluated inside the fused kernels,
csrc/engine_f32.hip.inc ``write_temb``)."""
import math

import torch
import torch.nn as nn

ENCODERS = {}


def register_encoder(module: nn.Module, name: str):
    """encoders.py:8-21"""
    global ENCODERS
    if name.lower() in ENCODERS:
        raise KeyError(f"Encoder {name} is already registered.")
    ENCODERS[name.lower()] = module


def get_encoder(type: str):
    """encoders.py:24-33"""
    if type.lower() in ENCODERS:
        return ENCODERS[type.lower()]
    raise ValueError(f"Unknown encoder type: {type}")


class GaussianFourierProjectionTimeEncoder(nn.Module):
    """phi(t) = [sin(t W), cos(t W)], W = randn(embed_dim/2) * scale * 2 pi, frozen by default
    (encoders.py:36-86)."""

    def __init__(self, embed_dim, scale=30.0, requires_grad=False):
        super().__init__()
        self.embed_dim = embed_dim
        self.W = nn.Parameter(torch.randn(embed_dim // 2) * scale * 2 * math.pi, requires_grad=requires_grad)

    def forward(self, x):
        raise NotImplementedError(
            "GaussianFourierProjectionTimeEncoder is evaluated inside the fused HIP kernels; call the "
            "owning IntrinsicModel(t, x, condition) instead of the encoder alone")


ENCODERS["GaussianFourierProjectionTimeEncoder".lower()] = GaussianFourierProjectionTimeEncoder
