"""QGPOCritic (grl/algorithms/qgpo.py:29-139): double Q critic + target copy, in-support softmax Q-learning loss.
Target values are HIP forwards of the target critics over the (B, K) fake next actions; the two regressions run as
fused forward+backward calls (``grlhip_loss_fwd_bwd`` on QNET plans) and expose their gradients to autograd."""
import copy

import torch
import torch.nn as nn

from ..config import as_config
from ..generative_models.model_functions.losses import fused_loss
from ..rl_modules.value_network import DoubleQNetwork


class QGPOCritic(nn.Module):
    def __init__(self, config):
        super().__init__()
        config = as_config(config)
        self.config = config
        self.q_alpha = config.q_alpha
        self.q = DoubleQNetwork(config.DoubleQNetwork)
        self.q_target = copy.deepcopy(self.q).requires_grad_(False)
        for net in self.q_target.model.values():   # the reference's soft update writes through ``.data``
            net.always_repack = True

    def forward(self, action, state=None):
        return self.q(action, state)

    def compute_double_q(self, action, state=None):
        return self.q.compute_double_q(action, state)

    def q_loss(self, action, state, reward, next_state, done, fake_next_action, discount_factor: float = 1.0):
        """:85-139"""
        with torch.no_grad():
            K = fake_next_action.shape[1]
            new_next_state = torch.stack([next_state] * K, dim=1)
            next_energy = self.q_target(fake_next_action, new_next_state).detach().squeeze(dim=-1)
            next_v = torch.sum(torch.softmax(self.q_alpha * next_energy, dim=1) * next_energy, dim=-1, keepdim=True)
            targets = reward + (1.0 - done.float()) * discount_factor * next_v
        B = action.shape[0]
        if targets.shape != (B, 1):
            # F.mse_loss broadcasts (B, 1) against other shapes; the fused regression is per-row only
            raise NotImplementedError("q_loss: reward/done must be (B, 1) tensors (got targets of shape %s)" % (tuple(targets.shape),))
        one = torch.ones(B, device=action.device)
        total = 0.0
        for net in (self.q.model["q1"], self.q.model["q2"]):
            net.set_io_dims(action.shape[-1], state.shape[-1])
            # fused row loss = 0.5 (q - target)^2, so its mean is half the MSE: (mse1 + mse2) / 2 = mean1 + mean2
            total = total + fused_loss(net, None, action, state, one, None, targets, average=True)
        return total
