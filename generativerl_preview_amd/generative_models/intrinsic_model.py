"""IntrinsicModel (grl/generative_models/intrinsic_model.py:12-75): t_encoder -> backbone(t, x[, c]),
built from config through the encoder / module registries.

The parameters live in ordinary ``nn.Parameter``s under the reference's names
(``model.t_encoder.W``, ``model.backbone.sort_t.0.weight`` ...).  ``forward`` evaluates the whole
network in the HIP library (``grlhip_net_forward``); the packed weight image is derived from the
parameters and refreshed whenever a parameter's version counter changes."""
import torch
import torch.nn as nn

from .. import _lib as L
from ..config import as_config
from ..engine import NetPlan, differentiable_forward
from ..neural_network import ConcatenateMLP, TemporalSpatialResidualNet, get_module
from ..neural_network.encoders import GaussianFourierProjectionTimeEncoder, get_encoder


class IntrinsicModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        config = as_config(config)
        self.config = config
        assert hasattr(config, "backbone"), "backbone must be specified in config"
        self.model = nn.ModuleDict()
        for enc in ("t_encoder", "x_encoder", "condition_encoder"):
            if hasattr(config, enc):
                self.model[enc] = get_encoder(config[enc].type)(**config[enc].args)
            else:
                self.model[enc] = nn.Identity()
        self.model["backbone"] = get_module(config.backbone.type)(**config.backbone.args)
        self.compute_dtype = L.F32
        self.backward_split = 3      # bf16 pieces per fp32 value in the backward GEMMs (NetPlan.set_backward_split)
        self._plan = None
        self._packed = {}

    # derived HIP state (plan handle, packed images, gradient staging) is never copied or pickled: a deep copy
    # (target networks) or a checkpoint rebuilds it lazily from the parameters
    _DERIVED = ("_plan", "_packed", "_flat_grads", "_grad_stage", "_stage_token")

    def __getstate__(self):
        st = dict(self.__dict__)
        for k in self._DERIVED:
            st.pop(k, None)
        st["_plan"], st["_packed"] = None, {}
        return st

    def _device(self):
        return next(self.model["backbone"].parameters()).device

    # ------------------------------------------------------------------ HIP plumbing
    @property
    def plan(self) -> NetPlan:
        dev = self._device()
        if self._plan is not None and dev.type == "cuda" and self._plan.device is not None and self._plan.device != dev:
            # the module moved to another GPU: the plan's program tables and every derived image live on the old one
            self._plan, self._packed = None, {}
            for k in ("_flat_grads", "_grad_stage", "_stage_token"):
                self.__dict__.pop(k, None)
        if self._plan is None:
            te, bb = self.model["t_encoder"], self.model["backbone"]
            if not isinstance(te, GaussianFourierProjectionTimeEncoder) or \
                    not isinstance(self.model["x_encoder"], nn.Identity) or \
                    not isinstance(self.model["condition_encoder"], nn.Identity):
                raise NotImplementedError("the fused kernels need a GaussianFourierProjectionTimeEncoder t_encoder "
                                          "and identity x/condition encoders")
            if isinstance(bb, TemporalSpatialResidualNet):
                d = bb.hip_desc()
                assert d["t_dim"] == te.embed_dim, "backbone t_dim must equal the t_encoder embed_dim"
                self._plan = NetPlan(L.NET_RESNET, d["x_dim"], d["out_dim"], d["cond_dim"], d["t_dim"],
                                     d["cond_hidden"], d["t_cond_hidden"], d["hidden"], device=dev)
            elif isinstance(bb, ConcatenateMLP):
                d = bb.hip_desc()
                # input is cat[t_emb, x, c]; x/c split is fixed at first use (set_io_dims)
                if not hasattr(self, "_io_dims"):
                    raise RuntimeError("ConcatenateMLP IntrinsicModel: call set_io_dims(x_dim, cond_dim) first")
                x_dim, c_dim = self._io_dims
                assert d["in_dim"] == te.embed_dim + x_dim + c_dim, "ConcatenateMLP input width mismatch"
                self._plan = NetPlan(L.NET_CONCAT_MLP, x_dim, d["out_dim"], c_dim, te.embed_dim, 0, 0, d["hidden"], device=dev)
            else:
                raise NotImplementedError("no fused kernel for backbone %s" % type(bb).__name__)
        # the Fourier frequencies are frozen unless the encoder was built with requires_grad=True (encoders.py:53-66):
        # the training entry points then skip the work that only feeds their gradient
        self._plan.set_fourier_grad(bool(self.model["t_encoder"].W.requires_grad))
        if self._plan.backward_split != getattr(self, "backward_split", 3):
            self._plan.set_backward_split(getattr(self, "backward_split", 3))
        return self._plan

    def set_io_dims(self, x_dim, cond_dim):
        self._io_dims = (int(x_dim), int(cond_dim or 0))

    def _named(self):
        sd = {k: v for k, v in self.model.named_parameters()}
        return {name: sd[name] for name, _ in self.plan.params}

    def packed(self, dtype=None):
        dtype = self.compute_dtype if dtype is None else dtype
        named = self._named()
        key = tuple((p.data_ptr(), p._version) for p in named.values())
        hit = self._packed.get(dtype)
        if hit is None or hit[0] != key:
            hit = (key, self.plan.pack(named, dtype))
            self._packed[dtype] = hit
        return hit[1]

    # ------------------------------------------------------------------ reference API
    def forward(self, t, x, condition=None):
        """intrinsic_model.py:54-75.  Differentiable with respect to the parameters (grlhip_net_backward recomputes
        the forward), not with respect to t / x / condition; the ``*_loss`` methods use fused single-call kernels
        instead."""
        if isinstance(self.model["backbone"], ConcatenateMLP) and not hasattr(self, "_io_dims"):
            self.set_io_dims(x.shape[-1], 0 if condition is None else condition.shape[-1])
        t = t.expand(x.shape[0]) if t.dim() == 0 else t
        return differentiable_forward(self, t, x, condition)
