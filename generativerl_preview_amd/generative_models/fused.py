"""Descriptors that let the solvers hand a whole integration to the fused HIP integrator.

The reference builds Python closures (``DiffusionProcess.reverse_ode(...).drift``) and lets a
third-party integrator call them once per step.  Every such closure has the form
    drift(t, x) = a(tau) x + b(tau) net(tau, x, c) + g(tau) dE/dx(tau, x, c),   tau = T - t (or t),
with scalar schedule coefficients, so here the closure is replaced by a ``FusedDrift`` that can
(1) tabulate (tau, dt, a, b, g, s) for a given time grid on the host and (2) name the packed
network(s) -- ``ODESolver`` / ``SDESolver`` pass both to ``grlhip_sample_ode`` / ``grlhip_sample_sde``.
Coefficient formulas: grl/generative_models/diffusion_process.py:742-919 (reverse SDE / ODE) with the
head conversions of model_functions/score_function.py:64-80 folded in for guided sampling."""
import torch

from .. import _lib as L


def _head_to_score(model_type, path, tau):
    """score = p * net + q * x  (ScoreFunction.forward, score_function.py:64-80)."""
    one, zero = torch.ones_like(tau), torch.zeros_like(tau)
    if model_type == "noise_function":
        return -one / path.std(tau), zero
    if model_type == "score_function":
        return one, zero
    if model_type == "velocity_function":
        g2 = path.diffusion_squared(tau)
        return -2.0 / g2, 2.0 * path.drift_coefficient(tau) / g2
    if model_type == "data_prediction_function":
        cov = path.covariance(tau)
        return path.scale(tau) / cov, -one / cov
    raise NotImplementedError("Unknown type of function {}".format(model_type))


class FusedDrift:
    """kind: 'reverse_ode' | 'reverse_sde' | 'icfm' | 'forward_ode' | 'forward_sde' | 'icfm_reverse'."""

    def __init__(self, kind, model, model_type, path=None, reverse_path=None, condition=None,
                 guidance=None, guidance_scale=0.0, T=1.0):
        self.kind, self.model, self.model_type = kind, model, model_type
        self.path, self.reverse_path, self.condition = path, reverse_path, condition
        self.guidance, self.guidance_scale, self.T = guidance, float(guidance_scale), T
        self.fixed = None   # (fixed_x, fixed_mask, noise | None): see with_fixed

    def with_fixed(self, fixed_x, fixed_mask, noise=None):
        """The same drift with some components pinned (diffusion_model.py:647-661): evaluated at
        ``direct_sample(T - t, fixed_x) * (1 - mask) + x * mask`` and multiplied by the mask.  ``noise``
        (n_steps, R, D) are the Gaussian draws of ``direct_sample``, one per solver step (drawn with ``torch.randn``
        on the device when None)."""
        import copy
        d = copy.copy(self)
        d.fixed = (fixed_x, fixed_mask, noise)
        return d

    # ---- per-step coefficients at solver times t (1-D fp32 CPU tensor) -> tau, a, b, g, s
    def coefficients(self, t):
        t = t.detach().to("cpu", torch.float32)
        zero = torch.zeros_like(t)
        if self.kind == "icfm":
            # drift = model(t, x, c), conditional_flow_model/independent_conditional_flow_model.py:285-288
            return t, zero, torch.ones_like(t), zero, zero
        if self.kind == "icfm_reverse":
            # drift = -model(t_max - t + t_min, x, c), independent_conditional_flow_model.py:807-809; T = (t_min, t_max)
            return (self.T[1] - t + self.T[0]).to(torch.float32), zero, -torch.ones_like(t), zero, zero
        if self.kind in ("forward_ode", "forward_sde"):
            # diffusion_process.py:559-740: the model is evaluated at the solver's own time
            p, mt = self.path, self.model_type
            f, g2 = p.drift_coefficient(t), p.diffusion_squared(t)
            if self.kind == "forward_sde":
                fp = self.reverse_path   # the process whose diffusion drives the forward SDE
                d2, s = g2 - fp.diffusion_squared(t), fp.diffusion(t)
                r = fp.diffusion_squared(t) / (g2 + 1e-8)
            else:
                d2, s, r = g2, zero, zero
            if mt == "score_function":
                return t, f, -0.5 * d2, zero, s
            if mt == "noise_function":
                return t, f, 0.5 * d2 / p.std(t), zero, s
            if mt == "velocity_function":
                return t, r * f, 1.0 - r, zero, s
            if mt == "data_prediction_function":
                D = 0.5 * d2 / p.covariance(t)
                return t, f + D, -p.scale(t) * D, zero, s
            raise NotImplementedError("Unknown type of function {}".format(mt))
        tau = torch.tensor(self.T, dtype=torch.float32) - t
        p, mt = self.path, self.model_type
        f, g2 = p.drift_coefficient(tau), p.diffusion_squared(tau)
        if self.kind == "reverse_sde":
            rp = self.reverse_path
            gr2, s = rp.diffusion_squared(tau), rp.diffusion(tau)
        else:
            gr2, s = None, zero
        if self.guidance is not None:
            # the guided score is integrated as a 'score_function' closure
            # (energy_conditional_diffusion_model.py:406-435,472-508): coefficient of the score ...
            cs = 0.5 * g2 if gr2 is None else 0.5 * (g2 + gr2)
            hp, hq = _head_to_score(mt, p, tau)
            return tau, -f + cs * hq, cs * hp, cs * self.guidance_scale, s
        if self.kind == "reverse_ode":  # diffusion_process.py:853-919
            if mt == "score_function":
                return tau, -f, 0.5 * g2, zero, s
            if mt == "noise_function":
                return tau, -f, -0.5 * g2 / p.std(tau), zero, s
            if mt == "velocity_function":
                return tau, zero, -torch.ones_like(t), zero, s
            if mt == "data_prediction_function":
                D = 0.5 * g2 / p.covariance(tau)
                return tau, -(f + D), p.scale(tau) * D, zero, s
        else:  # diffusion_process.py:742-851
            if mt == "score_function":
                return tau, -f, 0.5 * (g2 + gr2), zero, s
            if mt == "noise_function":
                return tau, -f, -0.5 * (g2 + gr2) / p.std(tau), zero, s
            if mt == "velocity_function":
                r = gr2 / (g2 + 1e-8)
                return tau, r * f, -(1.0 + r), zero, s
            if mt == "data_prediction_function":
                D = 0.5 * (g2 + gr2) / p.covariance(tau)
                return tau, -(f + D), p.scale(tau) * D, zero, s
        raise NotImplementedError("Unknown type of function {}".format(mt))

    def table(self, t_steps, dts, emit, lerp, emit2=None, lerp2=None):
        tau, a, b, g, s = self.coefficients(t_steps)
        n = len(t_steps)
        emit2 = torch.full((n,), -1.0) if emit2 is None else emit2.to(torch.float32)
        lerp2 = torch.ones(n) if lerp2 is None else lerp2.to(torch.float32)
        zero = torch.zeros(n)
        tab = torch.stack([tau, dts.to(torch.float32), a, b, g, s, emit.to(torch.float32), lerp.to(torch.float32),
                           emit2, lerp2, zero, zero], 1)
        assert tab.shape[1] == L.COEF_STRIDE
        return tab.contiguous()

    # ---- launch
    def integrate(self, x0, table, increments=None, trajectory_slots=0, dtype=None):
        m = self.model
        dtype = m.compute_dtype if dtype is None else dtype
        g_plan = g_packed = None
        if self.fixed is not None:
            dtype = L.F32   # the batch engine (the fused bf16 integrator has no fixed-component form)
        if self.guidance is not None:
            g_plan, g_packed = self.guidance.plan, self.guidance.packed(L.F32 if dtype == L.F32 else dtype)
        if self.fixed is not None:
            if self.kind not in ("reverse_ode", "reverse_sde"):
                raise NotImplementedError("fixed-component sampling is defined for the reverse diffusion processes only")
            fx, mask, noise = self.fixed
            tau = table[:, 0].to("cpu", torch.float32)
            fix_sc = torch.stack([self.path.scale(tau), self.path.std(tau)], 1).to(torch.float32).contiguous()
            rows = x0.shape[0]
            flat = lambda v: v.reshape(rows, -1)   # noqa: E731
            if noise is None:
                noise = torch.randn(table.shape[0], rows, flat(x0).shape[1], device=x0.device, dtype=torch.float32)
            return m.plan.sample_fixed(m.packed(dtype), flat(x0), self.condition, table, flat(fx), flat(mask), fix_sc,
                                       fix_noise=noise.reshape(table.shape[0], rows, -1), dW=increments, guidance=g_plan,
                                       guidance_packed=g_packed, trajectory_slots=trajectory_slots, dtype=dtype)
        return m.plan.sample(m.packed(dtype), x0, self.condition, table, dW=increments, guidance=g_plan,
                             guidance_packed=g_packed, trajectory_slots=trajectory_slots, dtype=dtype)

    def integrate_with_grad(self, x0, table, increments=None):
        """Differentiable Euler chain (``with_grad=True``): the fp32 trajectory (n_steps + 1, R, D) with a grad_fn whose
        backward walks the chain in the library (grlhip_sample_ode_backward) -- gradients reach the model's parameters
        and x0, not the condition.  Every step of ``table`` must emit slot k + 1 with lerp 1.
        Energy-guided chains: the reference detaches the guidance term (``calculate_energy_guidance`` returns
        ``guidance.detach()``, energy_conditional_diffusion_model.py:87-92), so it enters the forward chain only; the
        backward is the unguided chain rule along the guided trajectory, with no gradient to the energy net."""
        if self.fixed is not None:
            raise NotImplementedError("with_grad=True is not implemented for sampling with fixed components")
        named = self.model._named()
        params = [named[n] for n, _ in self.model.plan.params]
        return _DifferentiableOde.apply(self, table, increments, x0, *params)

    # a FusedDrift is also callable like the reference closure (one fused launch of a single step
    # evaluation is not provided; the solvers never call it)
    def __call__(self, t, x):
        raise NotImplementedError("FusedDrift is consumed whole by ODESolver/SDESolver.integrate")


class _DifferentiableOde(torch.autograd.Function):
    @staticmethod
    def forward(ctx, drift, table, increments, x0, *params):
        m = drift.model
        n = table.shape[0]
        g_plan = g_packed = None
        if drift.guidance is not None:
            g_plan, g_packed = drift.guidance.plan, drift.guidance.packed(L.F32)
        _, traj = m.plan.sample(m.packed(L.F32), x0, drift.condition, table, dW=increments, guidance=g_plan,
                                guidance_packed=g_packed, trajectory_slots=n + 1, dtype=L.F32)
        ctx.drift, ctx.table = drift, table
        ctx.save_for_backward(traj)
        return traj

    @staticmethod
    def backward(ctx, grad_traj):
        (traj,) = ctx.saved_tensors
        m = ctx.drift.model
        plan = m.plan
        flat, gx0 = plan.sample_ode_backward(m.packed(L.F32), traj, ctx.drift.condition, ctx.table, grad_traj)
        named = m._named()
        grads = []
        for i, (name, shape) in enumerate(plan.params):
            p = named[name]
            if p.requires_grad:
                off = L.lib().grlhip_grad_offset(plan.handle, i)
                grads.append(flat[off:off + p.numel()].view(p.shape))
            else:
                grads.append(None)
        return (None, None, None, gx0 if ctx.needs_input_grad[3] else None, *grads)


class _DifferentiableStages(torch.autograd.Function):
    """DPMSolver stages with a grad_fn: forward = ONE grlhip_sample_stages call (fp32) that stores every stage's
    input state, backward = ONE grlhip_sample_stages_backward call.  Returns the states (n_stages + 1, R, D)."""

    @staticmethod
    def forward(ctx, fn, condition, table, x0, *params):
        m = fn.model
        n = table.shape[0]
        g_plan = g_packed = None
        if fn.guidance is not None:
            g_plan, g_packed = fn.guidance.plan, fn.guidance.packed(L.F32)
        every = table.clone()
        every[:, 9] = torch.arange(1, n + 1, dtype=torch.float32)   # emit slot i + 1 after every stage
        _, states = m.plan.sample_stages(m.packed(L.F32), x0, condition, every, guidance=g_plan, guidance_packed=g_packed,
                                         trajectory_slots=n + 1, dtype=L.F32)
        ctx.fn, ctx.condition, ctx.table = fn, condition, table
        ctx.save_for_backward(states)
        return states

    @staticmethod
    def backward(ctx, grad_states):
        (states,) = ctx.saved_tensors
        m = ctx.fn.model
        plan = m.plan
        flat, gx0 = plan.sample_stages_backward(m.packed(L.F32), states, ctx.condition, ctx.table, grad_states)
        named = m._named()
        grads = []
        for i, (name, shape) in enumerate(plan.params):
            p = named[name]
            if p.requires_grad:
                off = L.lib().grlhip_grad_offset(plan.handle, i)
                grads.append(flat[off:off + p.numel()].view(p.shape))
            else:
                grads.append(None)
        return (None, None, None, gx0 if ctx.needs_input_grad[3] else None, *grads)


def differentiable_stages(fn, condition, table, x0):
    """States (n_stages + 1, R, D) of the stage table run from x0, differentiable with respect to the model's
    parameters and x0 (not the condition; the guidance term is a constant, as in the reference)."""
    named = fn.model._named()
    params = [named[n] for n, _ in fn.model.plan.params]
    return _DifferentiableStages.apply(fn, condition, table, x0, *params)


class AffineModelFn:
    """A solver-facing model function that is affine in (net, x, dE/dx):
        m(t, x, c) = cn(t) net(t, x, c) + cx(t) x + cg(t) dE/dx(t, x, c).
    Every noise / data-prediction function the reference hands to DPMSolver has this form (head conversions
    of model_functions/*.py, guided variants energy_conditional_diffusion_model.py:1235-1306).  Callable like
    the reference closure (host-driven evaluation: HIP forward + torch algebra), and carrying the three scalars
    so that DPMSolver can fold whole stages into ``grlhip_sample_stages``."""

    def __init__(self, kind, model, model_type, path, guidance=None, guidance_scale=0.0, call=None):
        assert kind in ("noise", "data")
        self.kind, self.model, self.model_type, self.path = kind, model, model_type, path
        self.guidance, self.guidance_scale, self._call = guidance, float(guidance_scale), call

    def __call__(self, t, x, condition=None):
        return self._call(t, x, condition)

    def coefficients(self, t):
        """t: 0-dim fp32 CPU tensor -> (cn, cx, cg) as Python floats."""
        p, mt = self.path, self.model_type
        t = t.reshape(1).to(torch.float32)
        sig, sc, cov = p.std(t), p.scale(t), p.covariance(t)
        if self.guidance is not None:
            # score_g = hp net + hq x + scale dE/dx; noise_g = -std score_g; data_g = (-std x + cov score_g) / scale
            hp, hq = _head_to_score(mt, p, t)
            if self.kind == "noise":
                cn, cx, cg = -sig * hp, -sig * hq, -sig * self.guidance_scale
            else:
                cn, cx, cg = cov * hp / sc, (-sig + cov * hq) / sc, cov * self.guidance_scale / sc
            return float(cn), float(cx), float(cg)
        one, zero = torch.ones_like(t), torch.zeros_like(t)
        f, g2 = p.drift_coefficient(t), p.diffusion_squared(t)
        if self.kind == "noise":      # NoiseFunction.forward, noise_function.py:62-78
            cn, cx = {"noise_function": (one, zero), "score_function": (-sig, zero),
                      "velocity_function": (2.0 * sig / g2, -f * 2.0 * sig / g2),
                      "data_prediction_function": (-sc / sig, one / sig)}[mt]
        else:                         # DataPredictionFunction.forward, data_prediction_function.py:62-84 (as written)
            cn, cx = {"noise_function": (-sig / sc, one / sc), "score_function": (cov / sc, -sig / sc),
                      "velocity_function": (-2.0 * cov / g2 / sc, (f * 2.0 * cov / g2 + 1.0) / sc),
                      "data_prediction_function": (one, zero)}[mt]
        return float(cn), float(cx), 0.0
