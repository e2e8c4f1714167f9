"""DiffusionModel (grl/generative_models/diffusion_model/diffusion_model.py:24-1233): config schema,
``sample`` / ``sample_forward_process`` shape contract and the two training losses, running on the
fused HIP integrator / fused loss kernels."""
from typing import List, Tuple, Union

import torch
import torch.nn as nn

from ...config import as_config
from ...numerical_methods.numerical_solvers import get_solver
from ...numerical_methods.numerical_solvers.dpm_solver import DPMSolver
from ...numerical_methods.numerical_solvers.ode_solver import ODESolver
from ...numerical_methods.numerical_solvers.sde_solver import SDESolver
from ...numerical_methods.probability_path import GaussianConditionalProbabilityPath
from ..diffusion_process import DiffusionProcess
from ..intrinsic_model import IntrinsicModel
from ..model_functions import heads
from ..model_functions.losses import fused_loss, prepare_flow_matching, prepare_score_matching
from ..random_generator import gaussian_random_variable
from ..sampling import expand_inputs, reshape_output

MODEL_TYPES = ["score_function", "data_prediction_function", "velocity_function", "noise_function"]


class DiffusionModel(nn.Module):
    def __init__(self, config) -> None:
        super().__init__()
        config = as_config(config)
        self.config = config
        self.x_size = config.x_size
        self.device = config.device
        self.alpha = config.alpha if hasattr(config, "alpha") else 1.0
        self.gaussian_generator = gaussian_random_variable(config.x_size, config.device,
                                                           config.use_tree_tensor if hasattr(config, "use_tree_tensor") else False)
        self.path = GaussianConditionalProbabilityPath(config.path)
        self.reverse_path = GaussianConditionalProbabilityPath(config.reverse_path) if hasattr(config, "reverse_path") else None
        self.model_type = config.model.type
        assert self.model_type in MODEL_TYPES, "Unknown type of model {}".format(self.model_type)
        self.model = IntrinsicModel(config.model.args)
        self.diffusion_process = DiffusionProcess(self.path)
        self.reverse_diffusion_process = DiffusionProcess(self.reverse_path) if self.reverse_path is not None else None
        if hasattr(config, "solver"):
            self.solver = get_solver(config.solver.type)(**config.solver.args)

    def get_type(self):
        return "DiffusionModel"

    # ------------------------------------------------------------------ heads (diffusion_model.py:985-1123)
    def score_function(self, t, x, condition=None):
        return heads.score(self.model_type, self.path, self.model(t, x, condition), t, x)

    def noise_function(self, t, x, condition=None):
        return heads.noise(self.model_type, self.path, self.model(t, x, condition), t, x)

    def velocity_function(self, t, x, condition=None):
        return heads.velocity(self.model_type, self.path, self.model(t, x, condition), t, x)

    def data_prediction_function(self, t, x, condition=None):
        return heads.data_prediction(self.model_type, self.path, self.model(t, x, condition), t, x)

    # ------------------------------------------------------------------ sampling
    def sample(self, t_span: torch.Tensor = None, batch_size: Union[torch.Size, int, Tuple[int], List[int]] = None,
               x_0=None, condition=None, with_grad: bool = False, solver_config=None, **kw):
        """diffusion_model.py:94-133 -- final state of ``sample_forward_process``."""
        return self.sample_forward_process(t_span=t_span, batch_size=batch_size, x_0=x_0, condition=condition,
                                           with_grad=with_grad, solver_config=solver_config,
                                           _final_only=True, **kw)

    def _function(self, guidance_scale):
        return self.model, self.model_type

    def _dpm_functions(self, guidance_scale):
        from ..fused import AffineModelFn
        return (AffineModelFn("noise", self.model, self.model_type, self.path, call=self.noise_function),
                AffineModelFn("data", self.model, self.model_type, self.path, call=self.data_prediction_function))

    def sample_forward_process(self, t_span: torch.Tensor = None, batch_size=None, x_0=None, condition=None,
                               with_grad: bool = False, solver_config=None, noise_increments=None,
                               _final_only=False, _guidance_scale=None):
        """diffusion_model.py:135-470.  Returns (T, ...) with the reference's shape rules.
        ``noise_increments`` (superset) pins the Brownian path of the SDE solver."""
        if t_span is not None:
            t_span = t_span.to(self.device)
        extra, n, x, condition = expand_inputs(self, batch_size, x_0, condition)
        if solver_config is not None:
            solver_config = as_config(solver_config)
            solver = get_solver(solver_config.type)(**solver_config.args)
        else:
            assert hasattr(self, "solver"), "solver must be specified in config or solver_config"
            solver = self.solver
        function, function_type = self._function(_guidance_scale)
        if isinstance(solver, DPMSolver):
            # diffusion_model.py:255-279 / energy_conditional_diffusion_model.py:370-405
            assert t_span is None, "DPMSolver does not support t_span argument assignment"
            noise_fn, data_fn = self._dpm_functions(_guidance_scale)
            data = solver.integrate(diffusion_process=self.diffusion_process, noise_function=noise_fn,
                                    data_prediction_function=data_fn, x=x, condition=condition, save_intermediate=True,
                                    with_grad=with_grad)
            if _final_only:
                data = data[-1]
        elif isinstance(solver, ODESolver):
            drift = self.diffusion_process.reverse_ode(function=function, function_type=function_type,
                                                       condition=condition).drift
            data = solver.integrate(drift=drift, x0=x, t_span=t_span, return_trajectory=not _final_only, with_grad=with_grad)
        elif isinstance(solver, SDESolver):
            assert self.reverse_diffusion_process is not None, "reverse_path must be specified in config"
            sde = self.diffusion_process.reverse_sde(
                function=function, function_type=function_type, condition=condition,
                reverse_diffusion_function=self.reverse_diffusion_process.diffusion,
                reverse_diffusion_squared_function=self.reverse_diffusion_process.diffusion_squared)
            data = solver.integrate(drift=sde.drift, diffusion=sde.diffusion, x0=x, t_span=t_span,
                                    noise_increments=noise_increments, return_trajectory=not _final_only,
                                    with_grad=with_grad)
        else:
            raise NotImplementedError("Solver type {} is not implemented".format(type(solver).__name__))
        if _final_only:
            data = data.unsqueeze(0)
        data = reshape_output(self, data, batch_size, extra, n, x_0, condition)
        return data[-1] if _final_only else data

    # ------------------------------------------------------------------ sampling with pinned components
    def sample_with_fixed_x(self, fixed_x, fixed_mask, t_span: torch.Tensor = None, batch_size=None, x_0=None,
                            condition=None, with_grad: bool = False, solver_config=None, **kw):
        """diffusion_model.py:472-519 -- final state of ``sample_forward_process_with_fixed_x``."""
        return self.sample_forward_process_with_fixed_x(fixed_x=fixed_x, fixed_mask=fixed_mask, t_span=t_span,
                                                        batch_size=batch_size, x_0=x_0, condition=condition,
                                                        with_grad=with_grad, solver_config=solver_config,
                                                        _final_only=True, **kw)

    def sample_forward_process_with_fixed_x(self, fixed_x, fixed_mask, t_span: torch.Tensor = None, batch_size=None,
                                            x_0=None, condition=None, with_grad: bool = False, solver_config=None,
                                            fixed_noise=None, noise_increments=None, _final_only=False,
                                            _guidance_scale=None):
        """diffusion_model.py:521-859: sampling with some components pinned to ``fixed_x`` (``fixed_mask`` is 1 where
        a component is generated, 0 where it is pinned).  The whole Euler / Euler-Maruyama loop is one library call
        (grlhip_sample_fixed, fp32 batch engine).  Returns (T, N, D), or (T, *B, N, D) with ``batch_size``.
        ``fixed_noise`` (superset, (n_steps, B*N, D)) pins the Gaussian draws that the reference makes inside the drift
        (``direct_sample``, one per solver step); ``noise_increments`` pins the Brownian path of the SDE solver."""
        if t_span is not None:
            t_span = t_span.to(self.device)
        if batch_size is None:
            extra = (1,)
        elif isinstance(batch_size, int):
            extra = (batch_size,)
        elif isinstance(batch_size, (torch.Size, Tuple, List)):
            extra = tuple(int(b) for b in batch_size)
        else:
            assert False, "Invalid batch size"
        n = fixed_x.shape[0]
        assert fixed_x.shape[0] == fixed_mask.shape[0], "The batch size of fixed_x and fixed_mask must be the same"
        if x_0 is not None and condition is not None:
            assert x_0.shape[0] == condition.shape[0], "The batch size of x_0 and condition must be the same"
        if x_0 is not None:
            assert x_0.shape[0] == n, "The batch size of x_0 and fixed_x must be the same"
        elif condition is not None:
            assert condition.shape[0] == n, "The batch size of condition and fixed_x must be the same"
        if solver_config is not None:
            solver_config = as_config(solver_config)
            solver = get_solver(solver_config.type)(**solver_config.args)
        else:
            assert hasattr(self, "solver"), "solver must be specified in config or solver_config"
            solver = self.solver
        rep = 1
        for b in extra:
            rep *= b
        if x_0 is None:
            x = self.gaussian_generator(batch_size=rep * n)
        else:
            xs = torch.Size([self.x_size]) if isinstance(self.x_size, int) else torch.Size(self.x_size)
            assert xs == x_0[0].shape, "The shape of x_0 must be the same as the x_size that is specified in the config"
            x = torch.repeat_interleave(x_0, rep, dim=0)
        if condition is not None:
            condition = torch.repeat_interleave(condition, rep, dim=0)
        fixed_x = torch.repeat_interleave(fixed_x.to(self.device, torch.float32), rep, dim=0)
        fixed_mask = torch.repeat_interleave(fixed_mask.to(self.device, torch.float32), rep, dim=0)
        function, function_type = self._function(_guidance_scale)
        if isinstance(solver, DPMSolver):
            assert False, "Not implemented"   # diffusion_model.py:641-643
        elif isinstance(solver, ODESolver):
            drift = self.diffusion_process.reverse_ode(function=function, function_type=function_type,
                                                       condition=condition).drift.with_fixed(fixed_x, fixed_mask, fixed_noise)
            data = solver.integrate(drift=drift, x0=x, t_span=t_span, return_trajectory=not _final_only, with_grad=with_grad)
        elif isinstance(solver, SDESolver):
            assert self.reverse_diffusion_process is not None, "reverse_path must be specified in config"
            sde = self.diffusion_process.reverse_sde(
                function=function, function_type=function_type, condition=condition,
                reverse_diffusion_function=self.reverse_diffusion_process.diffusion,
                reverse_diffusion_squared_function=self.reverse_diffusion_process.diffusion_squared)
            d = sde.drift.with_fixed(fixed_x, fixed_mask, fixed_noise)
            data = solver.integrate(drift=d, diffusion=d, x0=x, t_span=t_span, noise_increments=noise_increments,
                                    return_trajectory=not _final_only, with_grad=with_grad)
        else:
            raise NotImplementedError("Solver type {} is not implemented".format(type(solver).__name__))
        if _final_only:
            data = data.unsqueeze(0)
        xs = (self.x_size,) if isinstance(self.x_size, int) else tuple(self.x_size)
        data = data.reshape(data.shape[0], *extra, n, *xs)
        if batch_size is None:
            data = data.squeeze(1)
        return data[-1] if _final_only else data

    # ------------------------------------------------------------------ encoding (data -> latent)
    def forward_sample(self, x, t_span: torch.Tensor, condition=None, with_grad: bool = False, solver_config=None, **kw):
        """diffusion_model.py:861-888 -- final state of ``forward_sample_process``."""
        return self.forward_sample_process(x=x, t_span=t_span, condition=condition, with_grad=with_grad,
                                           solver_config=solver_config, _final_only=True, **kw)

    def forward_sample_process(self, x, t_span: torch.Tensor, condition=None, with_grad: bool = False,
                               solver_config=None, noise_increments=None, _final_only=False):
        """diffusion_model.py:890-983: the forward (data -> latent) probability-flow ODE / forward SDE of the model,
        integrated on ``t_span`` by the fused integrator (same kernels as sampling: only the per-step coefficients
        differ, diffusion_process.py:559-740).  Returns (T, N, D) as the reference does (no batch bookkeeping)."""
        t_span = t_span.to(self.device)
        if solver_config is not None:
            solver_config = as_config(solver_config)
            solver = get_solver(solver_config.type)(**solver_config.args)
        else:
            assert hasattr(self, "solver"), "solver must be specified in config or solver_config"
            solver = self.solver
        if isinstance(solver, ODESolver):
            drift = self.diffusion_process.forward_ode(function=self.model, function_type=self.model_type,
                                                       condition=condition).drift
            return solver.integrate(drift=drift, x0=x, t_span=t_span, return_trajectory=not _final_only, with_grad=with_grad)
        if isinstance(solver, SDESolver):
            assert self.diffusion_process is not None, "path must be specified in config"
            sde = self.diffusion_process.forward_sde(
                function=self.model, function_type=self.model_type, condition=condition,
                forward_diffusion_function=self.diffusion_process.diffusion,
                forward_diffusion_squared_function=self.diffusion_process.diffusion_squared)
            return solver.integrate(drift=sde.drift, diffusion=sde.diffusion, x0=x, t_span=t_span,
                                    noise_increments=noise_increments, return_trajectory=not _final_only,
                                    with_grad=with_grad)
        raise NotImplementedError("Solver type {} is not implemented".format(type(solver).__name__))

    # ------------------------------------------------------------------ likelihood
    def log_prob(self, x=None, condition=None, using_Hutchinson_trace_estimator: bool = True, with_grad: bool = False,
                 t=None, noise=None):
        """diffusion_model.py:1125-1159 -> metric.py:14-139.  ``t`` (the integration grid, default
        linspace(1e-3, 1, 1000)) and ``noise`` (the Hutchinson probes) are supersets for reproducibility."""
        if with_grad:
            raise NotImplementedError("log_prob(with_grad=True) needs second derivatives of the network")
        from ..metric import compute_likelihood
        return compute_likelihood(self, x, t=t, condition=condition,
                                  using_Hutchinson_trace_estimator=using_Hutchinson_trace_estimator, noise=noise)

    def sample_with_log_prob(self, t_span: torch.Tensor, batch_size=None, condition=None, with_grad: bool = False,
                             solver_config=None):
        """diffusion_model.py:1161-1189.  ``with_grad=True`` differentiates the SAMPLING part only: the likelihood of a
        differentiable sample needs second derivatives of the network, which the library does not have (``log_prob``
        raises for it), so the log-probability is returned without a grad_fn and a warning says so."""
        x = self.sample_forward_process(t_span=t_span, batch_size=batch_size, condition=condition, with_grad=with_grad,
                                        solver_config=solver_config)[-1]
        if with_grad:
            import warnings
            warnings.warn("sample_with_log_prob(with_grad=True): the sample is differentiable, its log-probability is "
                          "computed without gradient (INTEGRATION.md, known deviations)")
        return x, self.log_prob(x=x.detach() if with_grad else x, condition=condition, with_grad=False)

    # ------------------------------------------------------------------ training
    def dpo_loss(self, ref_dm, data, beta, t=None, noise=None):
        """diffusion_model.py:1191-1233 (marked in-development there): Direct Preference Optimisation on pairs
        ``data = [x_w | x_l]`` (the first two columns win, the rest lose -- as written) against a frozen reference
        model.  The four network evaluations are the differentiable HIP forward (gradients reach this model's
        parameters through grlhip_net_backward; the reference model's terms are detached as in the reference).
        Supersets ``t`` / ``noise`` pin the draws (made in the reference's order: noise first, then t)."""
        x_w, x_l = data[:, :2], data[:, 2:]
        if noise is None:
            noise = torch.randn_like(x_w)
        eps = 1e-5
        if t is None:
            t = torch.rand(x_w.shape[0], device=x_w.device) * (self.diffusion_process.t_max - eps) + eps
        dp = self.diffusion_process
        noisy_w = dp.scale(t, x_w) * x_w + dp.std(t, x_w) * noise
        noisy_l = dp.scale(t, x_l) * x_l + dp.std(t, x_l) * noise
        err = lambda net, x: (net(t, x) - noise).norm(dim=1, keepdim=True).pow(2)   # noqa: E731
        with torch.no_grad():
            ref_w, ref_l = err(ref_dm.model, noisy_w), err(ref_dm.model, noisy_l)
        w_diff = err(self.model, noisy_w) - ref_w
        l_diff = err(self.model, noisy_l) - ref_l
        inside = -1 * beta * (w_diff - l_diff)
        return -1 * torch.log(torch.sigmoid(inside) + eps).mean()

    def score_matching_loss(self, x, condition=None, weighting_scheme: str = None, average: bool = True,
                            t=None, noise=None):
        """diffusion_model.py:1006-1041 -> score_function.py:83-251.  ``t`` / ``noise`` (superset)
        replace the internal draws."""
        t, x_t, alpha, beta, target = prepare_score_matching(self, self.model_type, self.diffusion_process, x,
                                                             self.gaussian_generator, weighting_scheme, t, noise)
        return fused_loss(self.model, t, x_t, condition, alpha, beta, target, average)

    def flow_matching_loss(self, x, condition=None, average: bool = True, t=None, noise=None):
        """diffusion_model.py:1064-1081 -> velocity_function.py:81-218."""
        t, x_t, alpha, beta, target = prepare_flow_matching(self.model_type, self.diffusion_process, x,
                                                            self.gaussian_generator, t, noise)
        return fused_loss(self.model, t, x_t, condition, alpha, beta, target, average)
