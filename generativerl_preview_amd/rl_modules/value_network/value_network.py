"""VNetwork (grl/rl_modules/value_network/value_network.py:12-70): MultiLayerPerceptron(ReLU) value function over the
state.  Runs as a QNET plan with an empty second input (``grlhip_net_forward`` / ``grlhip_net_backward``, fp32 batch
engine); parameters keep the reference's names (``model.backbone.model.linear0.weight`` ...)."""
import torch
import torch.nn as nn

from ... import _lib as L
from ...config import as_config
from ...engine import NetPlan, differentiable_forward
from ...neural_network import MultiLayerPerceptron, get_module


class VNetwork(nn.Module):
    def __init__(self, config):
        super().__init__()
        config = as_config(config)
        self.config = config
        if hasattr(config, "state_encoder") or hasattr(config, "condition_encoder"):
            raise NotImplementedError("the fused value-network kernels need identity state/condition encoders")
        self.model = nn.ModuleDict()
        self.model["state_encoder"] = nn.Identity()
        self.model["condition_encoder"] = nn.Identity()
        self.model["backbone"] = get_module(config.backbone.type)(**config.backbone.args)
        self._plan = None
        self._packed = None

    def __getstate__(self):   # the plan handle and packed image are derived state: never copied or pickled
        st = dict(self.__dict__)
        st["_plan"], st["_packed"] = None, None
        st.pop("_flat_grads", None); st.pop("_grad_stage", None); st.pop("_stage_token", None)
        return st

    @property
    def plan(self) -> NetPlan:
        if self._plan is None:
            bb = self.model["backbone"]
            if not isinstance(bb, MultiLayerPerceptron) or not bb._relu_fusable:
                raise NotImplementedError("the fused value-network kernels support a ReLU MultiLayerPerceptron backbone "
                                          "without dropout/layernorm/final activation/scale")
            sizes, out = bb._sizes
            self._plan = NetPlan(L.NET_QNET, sizes[0], out, 0, 0, 0, 0, sizes[1:], device=next(self.parameters()).device)
        return self._plan

    def _named(self):
        # plan names are ConcatenateMLP-relative ("backbone.model.model.linearN"); the bare MLP has one level less
        sd = {k: v for k, v in self.model.named_parameters()}
        return {name: sd[name.replace("backbone.model.model.", "backbone.model.")] for name, _ in self.plan.params}

    def packed(self, dtype=L.F32):
        named = self._named()
        key = tuple((p.data_ptr(), p._version) for p in named.values())
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, self.plan.pack(named, L.F32))
        return self._packed[1]

    def forward(self, state, condition=None):
        """:48-70"""
        if condition is not None:
            raise NotImplementedError("VNetwork with a condition input has no fused kernel")
        if not torch.is_tensor(state):
            raise NotImplementedError("dict-typed states are out of scope")
        lead = state.shape[:-1]
        return differentiable_forward(self, None, state.reshape(-1, state.shape[-1]), None).reshape(*lead, -1)
