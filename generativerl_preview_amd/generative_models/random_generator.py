"""gaussian_random_variable (grl/generative_models/random_generator.py:45-122,203-212), tensor branch:
returns a callable ``(batch_size=None) -> randn(*batch, *x_size)`` on the configured device."""
from typing import List, Tuple, Union

import torch


def gaussian_random_variable(data_size, device=torch.device("cpu"), use_tree_tensor: bool = False):
    if isinstance(data_size, dict):
        raise NotImplementedError("dict-typed states are outside the fused hot path")
    if isinstance(data_size, int):
        size = (data_size,)
    elif isinstance(data_size, (list, tuple, torch.Size)):
        size = tuple(data_size)
    else:
        raise ValueError("Invalid data size: {}".format(data_size))

    def generator(batch_size: Union[int, torch.Size, torch.Tensor, List[int], Tuple[int]] = None):
        if batch_size is None:
            batch = ()
        elif isinstance(batch_size, torch.Tensor):
            batch = (int(batch_size.item()),) if batch_size.dim() == 0 else tuple(int(b) for b in batch_size)
        elif isinstance(batch_size, int):
            batch = (batch_size,)
        else:
            batch = tuple(batch_size)
        return torch.randn(*batch, *size, device=device)

    return generator
