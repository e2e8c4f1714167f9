"""name -> nn.Module table (grl/neural_network/activation.py:37-53)."""
import torch.nn as nn

ACTIVATIONS = {
    "mish": nn.Mish, "tanh": nn.Tanh, "relu": nn.ReLU, "softplus": nn.Softplus, "elu": nn.ELU,
    "silu": nn.SiLU, "swish": nn.SiLU, "square": None, "identity": nn.Identity, "sigmoid": nn.Sigmoid,
    "gelu": nn.GELU, "leaky_relu": nn.LeakyReLU,
}


def get_activation(name: str):
    name = name.lower()
    if name not in ACTIVATIONS or ACTIVATIONS[name] is None:
        raise ValueError("Unknown activation function {}".format(name))
    return ACTIVATIONS[name]()
