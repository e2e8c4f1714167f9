"""Golden vectors for the GMPO IQL critic.  Container-only script:
    python tests/golden/make_golden_iql.py        ->  tests/golden/iql.npz
The reference's own code here: DoubleQNetwork and VNetwork (+ ConcatenateMLP / MultiLayerPerceptron), loaded from
grl/rl_modules/value_network/{q_network,value_network}.py by file path (the package __init__ imports gym, absent
here), and torch autograd.  grl/algorithms/gmpo.py cannot be imported (torchrl, gym, d4rl absent): the few lines of
GMPOCritic.v_loss / iql_q_loss (:41-42, :100-114) are restated below around the reference networks -- the loss
composition is "parity unpinned", the network forwards and their gradients are pinned."""
import copy
import importlib.util
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_shims import EasyDict  # noqa: E402


def _load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, "/root/reference/grl/rl_modules/value_network/%s.py" % name)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


DoubleQNetwork, VNetwork = _load("q_network").DoubleQNetwork, _load("value_network").VNetwork


class GMPOCritic(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.q = DoubleQNetwork(config.DoubleQNetwork)
        self.q_target = copy.deepcopy(self.q).requires_grad_(False)
        self.v = VNetwork(config.VNetwork)

    def v_loss(self, state, action, next_state, tau):
        with torch.no_grad():
            target_q = self.q_target(action, state).detach()
            next_v = self.v(next_state).detach()
        adv = target_q - self.v(state)
        return torch.mean(torch.abs(tau - (adv < 0).float()) * adv ** 2), next_v

    def iql_q_loss(self, state, action, reward, done, next_v, discount):
        q_target = reward + (1.0 - done.float()) * discount * next_v.detach()
        qs = self.q.compute_double_q(action, state)
        return sum(torch.nn.functional.mse_loss(q, q_target) for q in qs) / len(qs), torch.mean(qs[0]), torch.mean(q_target)


out = {}
for name, A, S, H, B, tau, disc in (("small", 3, 5, [64, 32], 9, 0.7, 0.99), ("halfcheetah", 6, 17, [256, 256], 8, 0.9, 0.99)):
    mlp = lambda n_in: dict(hidden_sizes=[n_in] + H, output_size=1, activation="relu")  # noqa: E731
    cfg = EasyDict(q_alpha=1.0, DoubleQNetwork=dict(backbone=dict(type="ConcatenateMLP", args=mlp(A + S))),
                   VNetwork=dict(backbone=dict(type="MultiLayerPerceptron", args=mlp(S))))
    c = GMPOCritic(cfg)
    gen = torch.Generator().manual_seed(4321 + A)
    with torch.no_grad():
        for k, p in c.state_dict().items():
            p.copy_(torch.randn(p.shape, generator=gen) / (p.shape[1] if p.dim() == 2 else 4) ** 0.5)
    state, action = torch.randn(B, S, generator=gen), torch.randn(B, A, generator=gen)
    next_state, reward = torch.randn(B, S, generator=gen), torch.randn(B, 1, generator=gen)
    done = torch.rand(B, 1, generator=gen) < 0.3
    v_loss, next_v = c.v_loss(state, action, next_state, tau)
    v_loss.backward()
    q_loss, q_mean, tgt_mean = c.iql_q_loss(state, action, reward, done, next_v, disc)
    q_loss.backward()
    for k, v in c.state_dict().items():
        out[name + "/w/" + k] = v.numpy().copy()
    for k, p in c.named_parameters():
        if p.grad is not None:
            out[name + "/grad/" + k] = p.grad.numpy().copy()
    for k, v in dict(state=state, action=action, next_state=next_state, reward=reward, done=done, next_v=next_v,
                     v=c.v(state).detach(), v_loss=v_loss.detach(), q_loss=q_loss.detach(), q_mean=q_mean.detach(),
                     target_mean=tgt_mean.detach()).items():
        out[name + "/" + k] = v.numpy()
    out[name + "/tau"], out[name + "/discount"] = np.float32(tau), np.float32(disc)
    print(name, float(v_loss.detach()), float(q_loss.detach()), len([k for k in out if k.startswith(name + "/grad/")]))
np.savez_compressed(os.path.join(HERE, "iql.npz"), **out)
print("wrote iql.npz", sum(a.nbytes for a in out.values()) // 1024, "KiB")
