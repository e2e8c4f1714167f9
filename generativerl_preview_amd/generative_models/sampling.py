"""Batch-shape bookkeeping shared by the three model classes' ``sample_forward_process``
(grl/generative_models/diffusion_model/diffusion_model.py:160-234,369-416; same code in
energy_conditional_diffusion_model.py:257-349,514-561 and
conditional_flow_model/independent_conditional_flow_model.py:170-260,366-413)."""
from typing import List, Tuple

import torch


def expand_inputs(model, batch_size, x_0, condition):
    """-> (extra_batch_size tuple, data_batch_size, x (B*N, ...), condition (B*N, ...))."""
    if batch_size is None:
        extra = (1,)
    elif isinstance(batch_size, int):
        extra = (batch_size,)
    elif isinstance(batch_size, (torch.Size, Tuple, List)):
        extra = tuple(int(b) for b in batch_size)
    else:
        assert False, "Invalid batch size"
    if x_0 is not None and condition is not None:
        if type(x_0) == type(condition):
            assert x_0.shape[0] == condition.shape[0], "The batch size of x_0 and condition must be the same"
        n = x_0.shape[0]
    elif x_0 is not None:
        n = x_0.shape[0]
    elif condition is not None:
        n = condition.shape[0]
    else:
        n = 1
    rep = 1
    for b in extra:
        rep *= b
    if x_0 is None:
        x = model.gaussian_generator(batch_size=rep * n)
    else:
        if isinstance(model.x_size, int):
            assert torch.Size([model.x_size]) == x_0.shape[1:], \
                "The shape of x_0 must be the same as the x_size that is specified in the config"
        elif isinstance(model.x_size, (Tuple, List, torch.Size)):
            assert torch.Size(model.x_size) == x_0.shape[1:], \
                "The shape of x_0 must be the same as the x_size that is specified in the config"
        else:
            assert False, "Invalid x_size"
        x = torch.repeat_interleave(x_0, rep, dim=0)
    if condition is not None:
        if not isinstance(condition, torch.Tensor):
            raise NotImplementedError("Not implemented")
        condition = torch.repeat_interleave(condition, rep, dim=0)
    return extra, n, x, condition


def reshape_output(model, data, batch_size, extra, n, x_0, condition):
    """(T, B*N, D) -> (T, *B, N, D) + the reference's squeezes.  Rows were laid out n*B+b by
    repeat_interleave but are read here as b*N+n -- kept as written."""
    xs = (model.x_size,) if isinstance(model.x_size, int) else tuple(model.x_size)
    data = data.reshape(data.shape[0], *extra, n, *xs)
    if batch_size is None:
        data = data.squeeze(1).squeeze(1) if (x_0 is None and condition is None) else data.squeeze(1)
    elif x_0 is None and condition is None:
        data = data.squeeze(1 + len(extra))
    return data
