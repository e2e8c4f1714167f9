"""Golden vectors for QGPOCritic.q_loss.  Container-only script:
    python tests/golden/make_golden_qloss.py        ->  tests/golden/qloss.npz
What is the reference's own code here: the networks -- DoubleQNetwork / QNetwork / ConcatenateMLP(ReLU), loaded
from grl/rl_modules/value_network/q_network.py by file path (its package __init__ imports gym, absent here) -- and
torch autograd.  What is NOT: grl/algorithms/qgpo.py cannot be imported (torchrl, gym, d4rl are absent), so the ten
lines of QGPOCritic.q_loss (:107-139) are restated below around the reference networks; the loss composition is
therefore "parity unpinned" (stated in DESIGN.md), the network forwards and their gradients are pinned.
Weights: seeded normal draws scaled by 1/sqrt(fan_in), target critic drawn independently of the online one."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_shims import EasyDict  # noqa: E402
import copy  # noqa: E402
import importlib.util  # noqa: E402
import torch.nn as nn  # noqa: E402

_spec = importlib.util.spec_from_file_location("ref_q_network", "/root/reference/grl/rl_modules/value_network/q_network.py")
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
DoubleQNetwork = _mod.DoubleQNetwork


class QGPOCritic(nn.Module):
    """the composition of grl/algorithms/qgpo.py:47-51 and :107-139 around the reference's DoubleQNetwork"""

    def __init__(self, config):
        super().__init__()
        self.q_alpha = config.q_alpha
        self.q = DoubleQNetwork(config.DoubleQNetwork)
        self.q_target = copy.deepcopy(self.q).requires_grad_(False)

    def forward(self, action, state):
        return self.q(action, state)

    def q_loss(self, action, state, reward, next_state, done, fake_next_action, discount_factor=1.0):
        with torch.no_grad():
            new_next_state = torch.stack([next_state] * fake_next_action.shape[1], axis=1)
            next_energy = self.q_target(fake_next_action, new_next_state).detach().squeeze(dim=-1)
            next_v = torch.sum(nn.Softmax(dim=1)(self.q_alpha * next_energy) * next_energy, dim=-1, keepdim=True)
        targets = reward + (1.0 - done.float()) * discount_factor * next_v.detach()
        q0, q1 = self.q.compute_double_q(action, state)
        return (torch.nn.functional.mse_loss(q0, targets) + torch.nn.functional.mse_loss(q1, targets)) / 2

out = {}
# (name, action dim, state dim, hidden, B, K, q_alpha, discount)
for name, A, S, H, B, K, q_alpha, disc in (("small", 3, 5, [64, 32], 7, 4, 1.0, 0.99), ("halfcheetah", 6, 17, [256, 256], 6, 16, 3.0, 0.99)):
    cfg = EasyDict(q_alpha=q_alpha, DoubleQNetwork=dict(backbone=dict(type="ConcatenateMLP", args=dict(
        hidden_sizes=[A + S] + H, output_size=1, activation="relu"))))
    c = QGPOCritic(cfg)
    gen = torch.Generator().manual_seed(1234 + A)
    with torch.no_grad():
        for k, p in c.state_dict().items():
            fan_in = p.shape[1] if p.dim() == 2 else 4
            p.copy_(torch.randn(p.shape, generator=gen) / fan_in ** 0.5)
    action, state = torch.randn(B, A, generator=gen), torch.randn(B, S, generator=gen)
    reward, next_state = torch.randn(B, 1, generator=gen), torch.randn(B, S, generator=gen)
    done = (torch.rand(B, 1, generator=gen) < 0.3)
    fake = torch.randn(B, K, A, generator=gen)
    loss = c.q_loss(action, state, reward, next_state, done, fake, discount_factor=disc)
    loss.backward()
    for k, v in c.state_dict().items():
        out[name + "/w/" + k] = v.numpy().copy()
    for k, p in c.named_parameters():
        if p.grad is not None:
            out[name + "/grad/" + k] = p.grad.numpy().copy()
    for k, v in dict(action=action, state=state, reward=reward, next_state=next_state, done=done, fake_next_action=fake).items():
        out[name + "/" + k] = v.numpy()
    out[name + "/loss"], out[name + "/q_alpha"], out[name + "/discount"] = loss.detach().numpy(), np.float32(q_alpha), np.float32(disc)
    out[name + "/min_q"] = c(action, state).detach().numpy()
    print(name, float(loss), sorted(k for k in out if k.startswith(name + "/grad/"))[:2], len([k for k in out if k.startswith(name + "/grad/")]))
np.savez_compressed(os.path.join(HERE, "qloss.npz"), **out)
print("wrote qloss.npz", sum(a.nbytes for a in out.values()) // 1024, "KiB")
