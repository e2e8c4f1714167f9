"""Module registry and the MLP backbones of the hot path
(grl/neural_network/__init__.py:18-36,189-342,475-583,716-736).

These classes declare the parameters with the reference's names/shapes (so reference checkpoints
load with ``load_state_dict``) and describe themselves to the HIP library (``hip_desc``); they are
evaluated by the fused kernels through ``IntrinsicModel`` -- their own ``forward`` is not a compute
path."""
from typing import List, Union

import torch
import torch.nn as nn

from .activation import get_activation
from .encoders import ENCODERS, GaussianFourierProjectionTimeEncoder, get_encoder, register_encoder  # noqa: F401

MODULES = {}


def register_module(module: nn.Module, name: str):
    """grl/neural_network/__init__.py:18-29"""
    global MODULES
    if name.lower() in MODULES:
        raise KeyError(f"Module {name} is already registered.")
    MODULES[name.lower()] = module


def get_module(type: str):
    """grl/neural_network/__init__.py:32-36"""
    if type.lower() in MODULES:
        return MODULES[type.lower()]
    raise ValueError(f"Unknown module type: {type}")


def _not_a_compute_path(self, *a, **k):
    raise NotImplementedError(
        "%s is evaluated by the fused HIP kernels through IntrinsicModel(t, x, condition); "
        "calling the backbone on its own is not supported" % type(self).__name__)


class TemporalSpatialResBlock(nn.Module):
    """grl/neural_network/__init__.py:189-234: h1 = act(dense1 x) + time_mlp(act(t)); out = act(dense2 h1) + modify_x(x)."""

    def __init__(self, input_dim, output_dim, t_dim=128, activation=None):
        super().__init__()
        activation = activation if activation is not None else nn.SiLU()
        self.time_mlp = nn.Sequential(activation, nn.Linear(t_dim, output_dim))
        self.dense1 = nn.Sequential(nn.Linear(input_dim, output_dim), activation)
        self.dense2 = nn.Sequential(nn.Linear(output_dim, output_dim), activation)
        self.modify_x = nn.Linear(input_dim, output_dim) if input_dim != output_dim else nn.Identity()

    forward = _not_a_compute_path


class TemporalSpatialResidualNet(nn.Module):
    """grl/neural_network/__init__.py:237-342 (constructor arguments and parameter names kept)."""

    def __init__(self, hidden_sizes: List[int], output_dim: int, t_dim: int, input_dim: int = None,
                 condition_dim: int = None, condition_hidden_dim: int = None, t_condition_hidden_dim: int = None):
        super().__init__()
        if input_dim is None:
            input_dim = output_dim
        if condition_dim is None or condition_dim <= 0:
            condition_dim = 0
            condition_hidden_dim = 0
            t_condition_dim = t_dim
            t_condition_hidden_dim = t_dim
        else:
            t_condition_dim = t_dim + condition_hidden_dim
            t_condition_hidden_dim = t_condition_hidden_dim if t_condition_hidden_dim is not None else t_condition_dim
            self.pre_sort_condition = nn.Sequential(nn.Linear(condition_dim, condition_hidden_dim), nn.SiLU())
        self.sort_t = nn.Sequential(nn.Linear(t_condition_dim, t_condition_hidden_dim), nn.SiLU(),
                                    nn.Linear(t_condition_hidden_dim, t_condition_hidden_dim))
        tch = t_condition_hidden_dim
        self.first_block = TemporalSpatialResBlock(input_dim, hidden_sizes[0], t_dim=tch)
        self.down_block = nn.ModuleList([TemporalSpatialResBlock(hidden_sizes[i], hidden_sizes[i + 1], t_dim=tch)
                                         for i in range(len(hidden_sizes) - 1)])
        self.middle_block = TemporalSpatialResBlock(hidden_sizes[-1], hidden_sizes[-1], t_dim=tch)
        self.up_block = nn.ModuleList([TemporalSpatialResBlock(hidden_sizes[i], hidden_sizes[i], t_dim=tch)
                                       for i in range(len(hidden_sizes) - 2, -1, -1)])
        self.last_block = nn.Linear(hidden_sizes[0] * 2, output_dim)
        self._hip = dict(kind="resnet", x_dim=input_dim, out_dim=output_dim, cond_dim=condition_dim, t_dim=t_dim,
                         cond_hidden=condition_hidden_dim, t_cond_hidden=t_condition_hidden_dim,
                         hidden=list(hidden_sizes))

    def hip_desc(self):
        return dict(self._hip)

    forward = _not_a_compute_path


class MultiLayerPerceptron(nn.Module):
    """grl/neural_network/__init__.py:475-555 (linear{i}/activation{i} naming kept)."""

    def __init__(self, hidden_sizes: List[int], output_size: int, activation: Union[str, List[str]],
                 dropout: float = None, layernorm: bool = False, final_activation: str = None,
                 scale: float = None, shrink: float = None):
        super().__init__()
        self.model = nn.Sequential()
        for i in range(len(hidden_sizes) - 1):
            self.model.add_module("linear" + str(i), nn.Linear(hidden_sizes[i], hidden_sizes[i + 1]))
            act = activation[i] if isinstance(activation, list) else activation
            self.model.add_module("activation" + str(i), get_activation(act))
            if dropout is not None and dropout > 0:
                self.model.add_module("dropout", nn.Dropout(dropout))
            if layernorm:
                self.model.add_module("layernorm", nn.LayerNorm(hidden_sizes[i + 1]))
        self.model.add_module("linear" + str(len(hidden_sizes) - 1), nn.Linear(hidden_sizes[-1], output_size))
        if final_activation is not None:
            self.model.add_module("final_activation", get_activation(final_activation))
        self.scale = nn.Parameter(torch.tensor(scale), requires_grad=False) if scale is not None else 1.0
        if shrink is not None:
            (self.model[-2] if final_activation is not None else self.model[-1]).weight.data.normal_(0, shrink)
        acts = activation if isinstance(activation, list) else [activation] * (len(hidden_sizes) - 1)
        self._fusable = (all(a.lower() in ("silu", "swish") for a in acts) and not layernorm
                         and not (dropout is not None and dropout > 0) and final_activation is None and scale is None)
        self._relu_fusable = (all(a.lower() == "relu" for a in acts) and not layernorm
                              and not (dropout is not None and dropout > 0) and final_activation is None and scale is None)
        self._sizes = (list(hidden_sizes), output_size)

    forward = _not_a_compute_path


class ConcatenateMLP(nn.Module):
    """grl/neural_network/__init__.py:558-583: MLP over cat(inputs)."""

    def __init__(self, **kwargs):
        super().__init__()
        self.model = MultiLayerPerceptron(**kwargs)

    def hip_desc(self):
        if not self.model._fusable:
            raise NotImplementedError("the fused energy-guidance kernels support SiLU MLPs without "
                                      "dropout/layernorm/final activation/scale")
        sizes, out = self.model._sizes
        return dict(kind="concat_mlp", in_dim=sizes[0], hidden=sizes[1:], out_dim=out)

    def hip_desc_qnet(self):
        if not self.model._relu_fusable:
            raise NotImplementedError("the fused Q-network kernels support ReLU MLPs without "
                                      "dropout/layernorm/final activation/scale")
        sizes, out = self.model._sizes
        return dict(kind="qnet", in_dim=sizes[0], hidden=sizes[1:], out_dim=out)

    forward = _not_a_compute_path


MODULES.update({
    "TemporalSpatialResidualNet".lower(): TemporalSpatialResidualNet,
    "MultiLayerPerceptron".lower(): MultiLayerPerceptron,
    "ConcatenateMLP".lower(): ConcatenateMLP,
})
