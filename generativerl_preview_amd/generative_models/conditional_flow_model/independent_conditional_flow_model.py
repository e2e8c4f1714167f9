"""IndependentConditionalFlowModel (ICFM), grl/generative_models/conditional_flow_model/
independent_conditional_flow_model.py:27-897: velocity net, Euler ODE sampling with
drift = model(t, x, c) on time 0 -> 1, and flow_matching_loss(x0, x1, condition)."""
import torch
import torch.nn as nn

from ...config import as_config
from ...numerical_methods.numerical_solvers import get_solver
from ...numerical_methods.numerical_solvers.ode_solver import ODESolver
from ...numerical_methods.probability_path import ConditionalProbabilityPath
from ..fused import FusedDrift
from ..intrinsic_model import IntrinsicModel
from ..model_functions.losses import fused_loss, prepare_icfm
from ..random_generator import gaussian_random_variable
from ..sampling import expand_inputs, reshape_output
from ..stochastic_process import StochasticProcess


class IndependentConditionalFlowModel(nn.Module):
    def __init__(self, config) -> None:
        super().__init__()
        config = as_config(config)
        self.config = config
        self.x_size = config.x_size
        self.device = config.device
        self.gaussian_generator = gaussian_random_variable(config.x_size, config.device,
                                                           config.use_tree_tensor if hasattr(config, "use_tree_tensor") else False)
        self.path = ConditionalProbabilityPath(config.path)
        if hasattr(config, "reverse_path"):
            self.reverse_path = ConditionalProbabilityPath(config.reverse_path)
        else:
            self.reverse_path = None
        self.model_type = config.model.type
        assert self.model_type in ["velocity_function"], "Unknown type of model {}".format(self.model_type)
        self.model = IntrinsicModel(config.model.args)
        self.stochastic_process = StochasticProcess(self.path)
        if hasattr(config, "solver"):
            self.solver = get_solver(config.solver.type)(**config.solver.args)

    def get_type(self):
        return "IndependentConditionalFlowModel"

    def sample(self, t_span=None, batch_size=None, x_0=None, condition=None, with_grad: bool = False,
               solver_config=None):
        """:80-119"""
        return self.sample_forward_process(t_span=t_span, batch_size=batch_size, x_0=x_0, condition=condition,
                                           with_grad=with_grad, solver_config=solver_config, _final_only=True)

    def sample_forward_process(self, t_span=None, batch_size=None, x_0=None, condition=None,
                               with_grad: bool = False, solver_config=None, _final_only=False):
        """:170-450"""
        if t_span is not None:
            t_span = t_span.to(self.device)
        extra, n, x, condition = expand_inputs(self, batch_size, x_0, condition)
        if solver_config is not None:
            solver_config = as_config(solver_config)
            solver = get_solver(solver_config.type)(**solver_config.args)
        else:
            assert hasattr(self, "solver"), "solver must be specified in config or solver_config"
            solver = self.solver
        if not isinstance(solver, ODESolver):
            raise NotImplementedError("Not implemented")  # :283-284,342-343
        drift = FusedDrift("icfm", self.model, self.model_type, condition=condition)
        data = solver.integrate(drift=drift, x0=x, t_span=t_span, return_trajectory=not _final_only, with_grad=with_grad)
        if _final_only:
            data = data.unsqueeze(0)
        data = reshape_output(self, data, batch_size, extra, n, x_0, condition)
        return data[-1] if _final_only else data

    def log_prob(self, x_1, log_prob_x_0=None, function_log_prob_x_0=None, condition=None, t=None,
                 using_Hutchinson_trace_estimator: bool = True, noise=None):
        """:515-631: log-likelihood of ``x_1`` -- Euler on ``t`` (default linspace(0, 1, 1000)) of the composite ODE
        d(x, logp)/dt = (f, -div f), f = -model(1 - t, x, c), then the prior's log-density at the reached x_0
        (``log_prob_x_0`` given, ``function_log_prob_x_0(x_0)``, or a standard normal).  The network forward and the
        vector-Jacobian products of the divergence are library calls (generative_models/metric.py); ``noise`` pins the
        Hutchinson probes."""
        from ..metric import integrate_with_divergence, standard_normal_log_prob
        drift = FusedDrift("icfm_reverse", self.model, self.model_type, condition=condition,
                           T=(torch.tensor(0.0), torch.tensor(1.0)))
        t_span = torch.linspace(0.0, 1.0, 1000) if t is None else t
        x0, diff = integrate_with_divergence(drift, x_1, t_span, condition, using_Hutchinson_trace_estimator, noise)
        if log_prob_x_0 is None:
            x0 = x0.reshape(x_1.shape)
            log_prob_x_0 = function_log_prob_x_0(x0) if function_log_prob_x_0 is not None else standard_normal_log_prob(x0)
        return log_prob_x_0 - diff

    def sample_with_log_prob(self, t_span=None, batch_size=None, x_0=None, condition=None, with_grad: bool = False,
                             solver_config=None, using_Hutchinson_trace_estimator: bool = True):
        """:633-676"""
        x = self.sample(t_span=t_span, batch_size=batch_size, x_0=x_0, condition=condition, with_grad=with_grad,
                        solver_config=solver_config)
        return x, self.log_prob(x_1=x, condition=condition, t=t_span,
                                using_Hutchinson_trace_estimator=using_Hutchinson_trace_estimator)

    def forward_sample(self, x, t_span, condition=None, with_grad: bool = False, solver_config=None):
        """:743-770 -- final state of ``forward_sample_process``."""
        return self.forward_sample_process(x=x, t_span=t_span, condition=condition, with_grad=with_grad,
                                           solver_config=solver_config, _final_only=True)

    def forward_sample_process(self, x, t_span, condition=None, with_grad: bool = False, solver_config=None,
                               _final_only=False):
        """:772-846: encode data to the latent space -- Euler on ``t_span`` with the drift
        ``-model(t_span.max() - t + t_span.min(), x, c)`` (:807-809); returns (T, N, D)."""
        t_span = t_span.to(self.device)
        if solver_config is not None:
            solver_config = as_config(solver_config)
            solver = get_solver(solver_config.type)(**solver_config.args)
        else:
            assert hasattr(self, "solver"), "solver must be specified in config or solver_config"
            solver = self.solver
        if not isinstance(solver, ODESolver):
            raise NotImplementedError("Solver type {} is not implemented".format(type(solver).__name__))
        ts = t_span.detach().to("cpu", torch.float32)
        drift = FusedDrift("icfm_reverse", self.model, self.model_type, condition=condition, T=(ts.min(), ts.max()))
        return solver.integrate(drift=drift, x0=x, t_span=t_span, return_trajectory=not _final_only, with_grad=with_grad)

    def flow_matching_loss(self, x0, x1, condition=None, average: bool = True, sum_all_elements: bool = True,
                           t=None, noise=None):
        """:678-697 -> velocity_function.py:220-302"""
        t, x_t, alpha, beta, target = prepare_icfm(self.stochastic_process, x0, x1, t, noise)
        if not sum_all_elements and not average:
            # element-wise loss (velocity_function.py:248-253): the differentiable HIP forward (grlhip_net_backward
            # behind autograd) and the squared error per element in torch
            return 0.5 * (self.model(t, x_t, condition) - target) ** 2
        return fused_loss(self.model, t, x_t, condition, alpha, beta, target, average)

    def flow_matching_loss_with_mask(self, x0, x1, condition=None, mask=None, average: bool = True, t=None, noise=None):
        """:699-741: the element-wise loss with the elements outside ``mask`` (bool, (B, D)) dropped, averaged per row
        over the kept elements (at least 1).  ``mask=None``: the ordinary loss (the reference's call on that branch
        passes the model as ``x0`` and raises)."""
        if mask is None:
            return self.flow_matching_loss(x0, x1, condition, average, t=t, noise=noise)
        loss = self.flow_matching_loss(x0, x1, condition, average=False, sum_all_elements=False, t=t, noise=noise)
        loss = loss.masked_fill(~mask, 0.0)
        num = torch.clamp(mask.sum(dim=tuple(range(1, mask.dim()))), min=1)
        loss = loss.sum(dim=tuple(range(1, loss.dim()))) / num
        return loss.mean() if average else loss

    def sample_with_mask(self, t_span=None, batch_size=None, x_0=None, condition=None, mask=None, x_1=None,
                         with_grad: bool = False, solver_config=None):
        """:121-168 -- final state of ``sample_with_mask_forward_process``."""
        return self.sample_with_mask_forward_process(t_span=t_span, batch_size=batch_size, x_0=x_0, condition=condition,
                                                     mask=mask, x_1=x_1, with_grad=with_grad,
                                                     solver_config=solver_config)[-1]

    def sample_with_mask_forward_process(self, t_span=None, batch_size=None, x_0=None, condition=None, mask=None,
                                         x_1=None, with_grad: bool = False, solver_config=None):
        """:452-513: the sampled trajectory with the elements under ``mask`` replaced by ``x_1``."""
        data = self.sample_forward_process(t_span=t_span, batch_size=batch_size, x_0=x_0, condition=condition,
                                           with_grad=with_grad, solver_config=solver_config)
        return data if mask is None else torch.where(mask, x_1, data)
