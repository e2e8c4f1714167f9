"""GMPOCritic / GMPOPolicy (grl/algorithms/gmpo.py:41-405): the IQL critic and the advantage-weighted policy losses of
GMPO.  The critics' forwards are HIP forwards that are differentiable with respect to
their parameters (``grlhip_net_backward``), so the expectile and regression losses are written with torch as in the
reference; the policy losses weight the per-row fused score/flow-matching loss (``average=False``), whose backward
re-runs the fused kernel with the weights as row weights."""
import copy

import torch
import torch.nn as nn

from ..config import as_config
from ..generative_models import DiffusionModel, IndependentConditionalFlowModel
from ..rl_modules.value_network import DoubleQNetwork, VNetwork


def asymmetric_l2_loss(u, tau):
    """:41-42"""
    return torch.mean(torch.abs(tau - (u < 0).float()) * u ** 2)


class GMPOCritic(nn.Module):
    def __init__(self, config):
        super().__init__()
        config = as_config(config)
        self.config = config
        self.q_alpha = config.q_alpha
        self.q = DoubleQNetwork(config.DoubleQNetwork)
        self.q_target = copy.deepcopy(self.q).requires_grad_(False)
        for net in self.q_target.model.values():   # the reference's soft update writes through ``.data``
            net.always_repack = True
        self.v = VNetwork(config.VNetwork)

    def forward(self, action, state=None):
        return self.q(action, state)

    def compute_double_q(self, action, state=None):
        return self.q.compute_double_q(action, state)

    def v_loss(self, state, action, next_state, tau):
        """:100-108"""
        with torch.no_grad():
            target_q = self.q_target(action, state).detach()
            next_v = self.v(next_state).detach()
        v = self.v(state)
        adv = target_q - v
        return asymmetric_l2_loss(adv, tau), next_v

    def iql_q_loss(self, state, action, reward, done, next_v, discount):
        """:110-114"""
        q_target = reward + (1.0 - done.float()) * discount * next_v.detach()
        qs = self.q.compute_double_q(action, state)
        q_loss = sum(torch.nn.functional.mse_loss(q, q_target) for q in qs) / len(qs)
        return q_loss, torch.mean(qs[0]), torch.mean(q_target)


class GMPOPolicy(nn.Module):
    def __init__(self, config):
        super().__init__()
        config = as_config(config)
        self.config = config
        self.device = config.device
        self.critic = GMPOCritic(config.critic)
        self.model_type = config.model_type
        if self.model_type == "DiffusionModel":
            self.base_model = DiffusionModel(config.model)
            self.guided_model = DiffusionModel(config.model)
            self.model_loss_type = config.model_loss_type
            assert self.model_loss_type in ["score_matching", "flow_matching"]
        elif self.model_type == "IndependentConditionalFlowModel":
            self.base_model = IndependentConditionalFlowModel(config.model)
            self.guided_model = IndependentConditionalFlowModel(config.model)
        else:  # OT-CFM / Schrodinger bridge models are out of scope
            raise NotImplementedError("GMPOPolicy model_type %s" % self.model_type)
        self.softmax = nn.Softmax(dim=1)

    def forward(self, state):
        return self.sample(state)

    def sample(self, state, batch_size=None, solver_config=None, t_span=None, with_grad: bool = False):
        """:169-194"""
        return self.guided_model.sample(t_span=t_span, condition=state, batch_size=batch_size, with_grad=with_grad,
                                        solver_config=solver_config)

    def behaviour_policy_sample(self, state, batch_size=None, solver_config=None, t_span=None, with_grad: bool = False):
        """:196-223"""
        return self.base_model.sample(t_span=t_span, condition=state, batch_size=batch_size, with_grad=with_grad,
                                      solver_config=solver_config)

    def compute_q(self, state, action):
        return self.critic(action, state)

    def _model_loss(self, model, action, state, maximum_likelihood, average):
        if self.model_type == "DiffusionModel":
            if self.model_loss_type == "score_matching":
                if maximum_likelihood:
                    return model.score_matching_loss(action, state, average=average)
                return model.score_matching_loss(action, state, weighting_scheme="vanilla", average=average)
            return model.flow_matching_loss(action, state, average=average)
        x0 = model.gaussian_generator(batch_size=state.shape[0])
        return model.flow_matching_loss(x0=x0, x1=action, condition=state, average=average)

    def behaviour_policy_loss(self, action, state, maximum_likelihood: bool = False):
        """:242-272"""
        return self._model_loss(self.base_model, action, state, maximum_likelihood, True)

    def policy_optimization_loss_by_advantage_weighted_regression(self, action, state, maximum_likelihood: bool = False,
                                                                  beta: float = 1.0, weight_clamp: float = 100.0):
        """:274-338"""
        model_loss = self._model_loss(self.guided_model, action, state, maximum_likelihood, False)
        with torch.no_grad():
            q_value = self.critic(action, state).squeeze(dim=-1)
            v_value = self.critic.v(state).squeeze(dim=-1)
            weight = torch.exp(beta * (q_value - v_value))
        clamped_weight = weight.clamp(max=weight_clamp)
        clamped_ratio = torch.mean((clamped_weight < weight).float())
        return torch.mean(model_loss * clamped_weight), torch.mean(weight), torch.mean(clamped_weight), clamped_ratio

    def policy_optimization_loss_by_advantage_weighted_regression_softmax(self, state, fake_action,
                                                                          maximum_likelihood: bool = False,
                                                                          beta: float = 1.0):
        """Softmax-weighted regression over each state's K support actions (reference :340-405).  Every (state,
        candidate) pair is one row of the fused per-row loss; the critic's Q over the same rows, soft-maxed per state
        at temperature 1 / beta, gives the weights.  The reference finishes with ``torch.mean(..., dim=1)`` of a 1-D
        tensor, which raises IndexError in torch; that call is kept so the failure is the same."""
        n_states, n_cand = fake_action.shape[0], fake_action.shape[1]
        rows_a = fake_action.flatten(0, 1)
        rows_s = state.unsqueeze(1).expand(n_states, n_cand, *state.shape[1:]).flatten(0, 1)
        with torch.no_grad():
            q = self.critic(rows_a, rows_s).view(n_states, n_cand)
        per_row = self._model_loss(self.guided_model, rows_a, rows_s, maximum_likelihood, False).view(n_states, n_cand)
        w = torch.softmax(beta * q, dim=1)
        loss = torch.mean((w * per_row).sum(dim=-1), dim=1)
        return loss, q.mean(), w.mean(), per_row.mean()
