"""Likelihood of samples under a generative model with a Gaussian prior (grl/generative_models/metric.py:14-139).

The reference integrates the composite ODE d(x, logp)/dt = (f(t, x), -div f(t, x)) from t = 1e-3 to 1 with Euler
(``ODESolver(library="torchdiffeq_adjoint")``, default ``ode_solver="euler"``), f = ``forward_ode`` drift of a
DiffusionModel or ``-model(1 - t, x, c)`` of a flow model, and the divergence either exactly (one vector-Jacobian product
per state dimension) or with the Hutchinson-Skilling estimator (one product with a Gaussian probe).

Here every f is affine in (x, net) with coefficients that depend on t only (FusedDrift.coefficients), so
div f = a(t) D + b(t) tr(d net / d x): the network forward is grlhip_net_forward and each vector-Jacobian product is the
input gradient of grlhip_net_backward -- two library calls per step (exact trace: 1 + D).  ``noise=`` (superset,
(n_steps, B, D)) pins the Hutchinson probes, which the reference draws inside the drift."""
import math

import torch

from .. import _lib as L
from .fused import FusedDrift


def _drift_of(model, condition):
    kind = model.get_type()
    if kind == "EnergyConditionalDiffusionModel":
        raise NotImplementedError("EnergyConditionalDiffusionModel is not supported yet.")   # metric.py:35-38
    if kind == "DiffusionModel":
        return model.diffusion_process.forward_ode(function=model.model, function_type=model.model_type,
                                                   condition=condition).drift
    if kind in ("IndependentConditionalFlowModel", "OptimalTransportConditionalFlowModel"):
        # -model(1 - t, x, c), metric.py:48
        return FusedDrift("icfm_reverse", model.model, model.model_type, condition=condition,
                          T=(torch.tensor(0.0), torch.tensor(1.0)))
    raise ValueError("Invalid model type: {}".format(kind))


def integrate_with_divergence(drift, x, t_span, condition=None, using_Hutchinson_trace_estimator: bool = True, noise=None):
    """Euler on ``t_span`` of d(x, logp)/dt = (f, -div f) for a FusedDrift f (the composite_drift of metric.py:86-106 /
    independent_conditional_flow_model.py:572-592).  Returns (x at t_span[-1] as (B, D), logp difference (B,))."""
    net = drift.model
    B = x.shape[0]
    dev = x.device
    x = x.detach().reshape(B, -1).to(torch.float32)
    D = x.shape[1]
    if not hasattr(net, "_io_dims"):
        net.set_io_dims(D, 0 if condition is None else condition.shape[-1])
    t_span = t_span.detach().to("cpu", torch.float32)
    n_steps = len(t_span) - 1
    tau, a, b, _, _ = drift.coefficients(t_span[:-1])
    dts = (t_span[1:] - t_span[:-1]).tolist()
    tau, a, b = tau.tolist(), a.tolist(), b.tolist()
    packed = net.packed(L.F32)
    plan = net.plan
    logp_diff = torch.zeros(B, device=dev, dtype=torch.float32)
    eye = torch.eye(D, device=dev, dtype=torch.float32)
    with torch.no_grad():
        for k in range(n_steps):
            tk = torch.full((B,), tau[k], device=dev, dtype=torch.float32)
            out = plan.forward(packed, tk, x, condition)
            if using_Hutchinson_trace_estimator:
                eps = noise[k].reshape(B, D).to(dev, torch.float32) if noise is not None else torch.randn_like(x)
                _, vjp = plan.backward(packed, tk, x, condition, eps, need_params=False, need_x=True)
                div = a[k] * (eps * eps).sum(1) + b[k] * (vjp * eps).sum(1)
            else:
                tr = torch.zeros(B, device=dev, dtype=torch.float32)
                for i in range(D):
                    _, vjp = plan.backward(packed, tk, x, condition, eye[i].expand(B, D).contiguous(), need_params=False,
                                           need_x=True)
                    tr += vjp[:, i]
                div = a[k] * D + b[k] * tr
            logp_diff = logp_diff + dts[k] * (-div)
            x = x + dts[k] * (a[k] * x + b[k] * out)
    return x, logp_diff


def standard_normal_log_prob(x):
    """Independent(Normal(0, 1), 1).log_prob of the flattened rows (metric.py:129-135)."""
    x = x.reshape(x.shape[0], -1)
    return -0.5 * (x * x).sum(1) - 0.5 * x.shape[1] * math.log(2.0 * math.pi)


def compute_likelihood(model, x, t=None, condition=None, using_Hutchinson_trace_estimator: bool = True, noise=None):
    """log p(x) of each row, shape (B,).  No gradient flows to the parameters (the reference's ``with_grad=True`` would
    need second derivatives of the network, which the library does not have)."""
    drift = _drift_of(model, condition)
    t_span = torch.linspace(1e-3, 1.0, 1000) if t is None else t
    x1, logp_diff = integrate_with_divergence(drift, x, t_span, condition, using_Hutchinson_trace_estimator, noise)
    return standard_normal_log_prob(x1) - logp_diff
