"""QNetwork / DoubleQNetwork (grl/rl_modules/value_network/q_network.py:12-133): ConcatenateMLP(ReLU) critics over
cat[action, state].  Parameters live in ordinary ``nn.Parameter``s under the reference's names
(``model.backbone.model.model.linear0.weight`` ...); ``forward`` evaluates the network in the HIP library
(``grlhip_net_forward`` on a QNET plan, fp32 batch engine) and is differentiable with respect to the parameters
(``grlhip_net_backward``); ``QGPOCritic.q_loss`` (algorithms/qgpo.py) uses one fused forward+backward call per critic."""
import copy

import torch
import torch.nn as nn

from ... import _lib as L
from ...config import as_config
from ...engine import NetPlan, differentiable_forward
from ...neural_network import ConcatenateMLP, get_module


class QNetwork(nn.Module):
    def __init__(self, config):
        super().__init__()
        config = as_config(config)
        self.config = config
        if hasattr(config, "action_encoder") or hasattr(config, "state_encoder"):
            raise NotImplementedError("the fused Q-network kernels need identity action/state encoders")
        self.model = nn.ModuleDict()
        self.model["action_encoder"] = nn.Identity()
        self.model["state_encoder"] = nn.Identity()
        self.model["backbone"] = get_module(config.backbone.type)(**config.backbone.args)
        self.always_repack = False   # set on target networks whose weights are updated through ``.data``
        self._plan = None
        self._io_dims = None
        self._packed = None

    def __deepcopy__(self, memo):
        twin = QNetwork(self.config)
        twin.load_state_dict(self.state_dict())
        for p, q in zip(twin.parameters(), self.parameters()):
            p.requires_grad_(q.requires_grad)
        twin.always_repack = self.always_repack
        return twin.to(next(self.parameters()).device)

    # ------------------------------------------------------------------ HIP plumbing
    def set_io_dims(self, action_dim, state_dim):
        dims = (int(action_dim), int(state_dim or 0))
        if self._io_dims is not None and self._io_dims != dims:
            raise ValueError("QNetwork was planned for (action, state) widths %s, got %s" % (self._io_dims, dims))
        self._io_dims = dims

    @property
    def plan(self) -> NetPlan:
        if self._plan is None:
            bb = self.model["backbone"]
            if not isinstance(bb, ConcatenateMLP):
                raise NotImplementedError("no fused kernel for Q backbone %s" % type(bb).__name__)
            d = bb.hip_desc_qnet()
            if self._io_dims is None:
                raise RuntimeError("QNetwork: call set_io_dims(action_dim, state_dim) first")
            a_dim, s_dim = self._io_dims
            assert d["in_dim"] == a_dim + s_dim, "ConcatenateMLP input width mismatch"
            self._plan = NetPlan(L.NET_QNET, a_dim, d["out_dim"], s_dim, 0, 0, 0, d["hidden"],
                                 device=next(self.parameters()).device)
        return self._plan

    def _named(self):
        sd = {k: v for k, v in self.model.named_parameters()}
        return {name: sd[name] for name, _ in self.plan.params}

    def packed(self, dtype=L.F32):
        named = self._named()
        key = tuple((p.data_ptr(), p._version) for p in named.values())
        if self._packed is None or self._packed[0] != key or self.always_repack:
            self._packed = (key, self.plan.pack(named, L.F32))
        return self._packed[1]

    # ------------------------------------------------------------------ reference API
    def forward(self, action, state):
        """:48-64.  action (..., A), state (..., S) -> (..., 1)"""
        if not torch.is_tensor(action) or not torch.is_tensor(state):
            raise NotImplementedError("dict-typed actions/states are out of scope")
        self.set_io_dims(action.shape[-1], state.shape[-1])
        lead = action.shape[:-1]
        a2 = action.reshape(-1, action.shape[-1])
        s2 = state.expand(*lead, state.shape[-1]).reshape(-1, state.shape[-1])
        return differentiable_forward(self, None, a2, s2).reshape(*lead, -1)


class DoubleQNetwork(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.model = nn.ModuleDict()
        self.model["q1"] = QNetwork(config)
        self.model["q2"] = QNetwork(config)

    def compute_double_q(self, action, state):
        return self.model["q1"](action, state), self.model["q2"](action, state)

    def compute_mininum_q(self, action, state):
        return torch.min(*self.compute_double_q(action, state))

    def forward(self, action, state):
        return self.compute_mininum_q(action, state)
