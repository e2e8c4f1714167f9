"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

A plain eager-PyTorch fp32 restatement of the reference's generative-policy hot path, written
from the reference's published behaviour (file:line cited per function, paths relative to
/root/reference).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this file; the product package (``generativerl_preview_amd``)
never does and fails loudly without its HIP library.

Parity status: every function here is pinned by ``tests/golden/*.npz`` (vectors produced by the
reference's own classes, see ``tests/golden/make_golden.py``) in ``tests/test_oracle_golden.py``
-- EXCEPT the time stepping itself (``euler_ode`` / ``euler_maruyama``): the reference delegates
that to torchdyn / torchdiffeq / torchsde, which are neither vendored, pinned nor installed, so
their fixed-grid Euler / dt-driven Euler-Maruyama semantics are restated from those libraries'
documented behaviour: **stepping parity unpinned** (SURVEY.md section 8c).

All tensors are float32 torch CPU tensors; ``sd`` is a mapping ``state_dict key -> tensor`` using
the reference's key layout (``model.model.backbone.sort_t.0.weight`` ...).
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- network
def fourier_embed(t, W):
    """grl/neural_network/encoders.py:85-86 -- cat[sin(t W), cos(t W)]."""
    p = t[..., None] * W[None, :]
    return torch.cat([torch.sin(p), torch.cos(p)], dim=-1)


def _lin(sd, key, x):
    return F.linear(x, sd[key + ".weight"], sd[key + ".bias"])


def _res_block(sd, pre, te, x):
    """grl/neural_network/__init__.py:224-234 (ctor :200-222)."""
    h1 = F.silu(_lin(sd, pre + "dense1.0", x)) + _lin(sd, pre + "time_mlp.1", F.silu(te))
    h2 = F.silu(_lin(sd, pre + "dense2.0", h1))
    if (pre + "modify_x.weight") in sd:
        return h2 + _lin(sd, pre + "modify_x", x)
    return h2 + x


def resnet_forward(sd, pre, t_emb, x, cond=None):
    """TemporalSpatialResidualNet.forward, grl/neural_network/__init__.py:329-342."""
    if cond is not None:
        tc = torch.cat([t_emb, F.silu(_lin(sd, pre + "pre_sort_condition.0", cond))], dim=-1)
    else:
        tc = t_emb
    te = _lin(sd, pre + "sort_t.2", F.silu(_lin(sd, pre + "sort_t.0", tc)))
    d = [_res_block(sd, pre + "first_block.", te, x)]
    i = 0
    while (pre + "down_block.%d.dense1.0.weight" % i) in sd:
        d.append(_res_block(sd, pre + "down_block.%d." % i, te, d[i]))
        i += 1
    u = _res_block(sd, pre + "middle_block.", te, d[-1])
    i = 0
    while (pre + "up_block.%d.dense1.0.weight" % i) in sd:
        u = _res_block(sd, pre + "up_block.%d." % i, te, torch.cat([u, d[-i - 1]], dim=-1))
        i += 1
    return _lin(sd, pre + "last_block", torch.cat([u, d[0]], dim=-1))


def concat_mlp_forward(sd, pre, *xs):
    """ConcatenateMLP / MultiLayerPerceptron with SiLU, grl/neural_network/__init__.py:548-583."""
    h = torch.cat(xs, dim=-1)
    i = 0
    while (pre + "model.model.linear%d.weight" % (i + 1)) in sd:
        h = F.silu(_lin(sd, pre + "model.model.linear%d" % i, h))
        i += 1
    return _lin(sd, pre + "model.model.linear%d" % i, h)


def intrinsic_forward(sd, t, x, cond=None, pre="model.", backbone="TemporalSpatialResidualNet"):
    """IntrinsicModel.forward, grl/generative_models/intrinsic_model.py:54-75
    (t_encoder = GaussianFourierProjectionTimeEncoder, identity x/condition encoders)."""
    te = fourier_embed(t, sd[pre + "model.t_encoder.W"])
    if backbone == "TemporalSpatialResidualNet":
        return resnet_forward(sd, pre + "model.backbone.", te, x, cond)
    if backbone == "ConcatenateMLP":
        ins = (te, x) if cond is None else (te, x, cond)
        return concat_mlp_forward(sd, pre + "model.backbone.", *ins)
    raise ValueError(backbone)


# ----------------------------------------------------------------------------- schedules
class Path:
    """GaussianConditionalProbabilityPath, grl/numerical_methods/probability_path.py:90-448
    (``linear_vp_sde`` and ``gvp``), reproduced as written -- including d_scale_dt's missing
    beta_0 term (:342-343) and gvp's g^2 (:222-229)."""

    def __init__(self, cfg):
        self.type = cfg["type"]
        self.b0 = float(cfg.get("beta_0", 0.1))
        self.b1 = float(cfg.get("beta_1", 20.0))
        self.t_max = float(cfg.get("t_max", 1.0))
        assert self.type in ("linear_vp_sde", "gvp"), self.type

    def f(self, t):
        if self.type == "linear_vp_sde":
            return -0.5 * (self.b0 + t * (self.b1 - self.b0))
        return -0.5 * torch.pi * torch.tan(torch.pi * t / 2.0)

    def g2(self, t):
        if self.type == "linear_vp_sde":
            return self.b0 + t * (self.b1 - self.b0)
        s = torch.sin(torch.pi * t * 0.5)
        return torch.pi * s + s * s * torch.tan(torch.pi * t * 0.5)

    def g(self, t):
        return torch.sqrt(self.g2(t))

    def scale(self, t):
        if self.type == "linear_vp_sde":
            return torch.exp(-0.25 * t ** 2 * (self.b1 - self.b0) - 0.5 * t * self.b0)
        return torch.cos(0.5 * torch.pi * t)

    def d_scale(self, t):
        if self.type == "linear_vp_sde":
            return -0.5 * t * (self.b1 - self.b0) * self.scale(t)
        return -0.5 * torch.pi * torch.sin(0.5 * torch.pi * t)

    def std(self, t):
        if self.type == "linear_vp_sde":
            return torch.sqrt(1.0 - self.scale(t) ** 2)
        return torch.sin(0.5 * torch.pi * t)

    def d_std(self, t):
        if self.type == "linear_vp_sde":
            return -self.d_scale(t) * self.scale(t) / self.std(t)
        return 0.5 * torch.pi * torch.cos(0.5 * torch.pi * t)

    def cov(self, t):
        if self.type == "linear_vp_sde":
            return 1.0 - self.scale(t) ** 2
        s = torch.sin(0.5 * torch.pi * t)
        return s * s

    def log_scale(self, t):  # probability_path.py:265-293
        if self.type == "linear_vp_sde":
            return -0.25 * t ** 2 * (self.b1 - self.b0) - 0.5 * t * self.b0
        return torch.log(torch.cos(0.5 * torch.pi * t))

    def half_log_snr(self, t):  # probability_path.py:479-498 (linear_vp_sde only, like the reference)
        assert self.type == "linear_vp_sde"
        return self.log_scale(t) - 0.5 * torch.log(1.0 - self.scale(t) ** 2)

    def inverse_half_log_snr(self, lam):  # probability_path.py:500-538
        assert self.type == "linear_vp_sde"
        num = 2.0 * torch.logaddexp(-2.0 * lam, torch.zeros((1,)).to(lam))
        den = torch.sqrt(self.b0 ** 2 + (self.b1 - self.b0) * num) + self.b0
        return num / den


def _b(v, x):
    """DiffusionProcess broadcast: (B,) -> x.shape, diffusion_process.py:55-57."""
    return v[(...,) + (None,) * (x.dim() - v.dim())].expand(x.shape) if x.dim() > v.dim() else v


# ----------------------------------------------------------------------------- head conversions
def score_head(model_type, path, net, t, x):
    """ScoreFunction.forward, model_functions/score_function.py:44-81."""
    if model_type == "noise_function":
        return -net / _b(path.std(t), x)
    if model_type == "score_function":
        return net
    if model_type == "velocity_function":
        return -(net - _b(path.f(t), x) * x) * 2.0 / _b(path.g2(t), x)
    if model_type == "data_prediction_function":
        return -(x - _b(path.scale(t), x) * net) / _b(path.cov(t), x)
    raise NotImplementedError(model_type)


def noise_head(model_type, path, net, t, x):
    """NoiseFunction.forward, model_functions/noise_function.py:62-78."""
    if model_type == "noise_function":
        return net
    if model_type == "score_function":
        return -net * _b(path.std(t), x)
    if model_type == "velocity_function":
        return (net - _b(path.f(t), x) * x) * 2.0 * _b(path.std(t), x) / _b(path.g2(t), x)
    if model_type == "data_prediction_function":
        return (x - _b(path.scale(t), x) * net) / _b(path.std(t), x)
    raise NotImplementedError(model_type)


def velocity_head(model_type, path, net, t, x):
    """VelocityFunction.forward, model_functions/velocity_function.py:54-79
    (the data_prediction branch raises in the reference -- shape bug -- and is not restated)."""
    if model_type == "noise_function":
        return _b(path.f(t), x) * x + 0.5 * _b(path.g2(t), x) / _b(path.std(t), x) * net
    if model_type == "score_function":
        return _b(path.f(t), x) * x - 0.5 * _b(path.g2(t), x) * net
    if model_type == "velocity_function":
        return net
    raise NotImplementedError(model_type)


def data_head(model_type, path, net, t, x):
    """DataPredictionFunction.forward, model_functions/data_prediction_function.py:62-84."""
    s, sig = _b(path.scale(t), x), _b(path.std(t), x)
    if model_type == "noise_function":
        return (x - sig * net) / s
    if model_type == "score_function":  # as written: -std * x (not x)
        return (-sig * x + _b(path.cov(t), x) * net) / s
    if model_type == "velocity_function":
        return ((_b(path.f(t), x) * x - net) * 2.0 * _b(path.cov(t), x) / _b(path.g2(t), x) + x) / s
    if model_type == "data_prediction_function":
        return net
    raise NotImplementedError(model_type)


# ----------------------------------------------------------------------------- drifts
def reverse_ode_drift(model_type, path, fn, t, x, T=1.0):
    """DiffusionProcess.reverse_ode, generative_models/diffusion_process.py:853-919.
    ``fn(tau, x)`` is the model (or the guided score when model_type == 'score_function')."""
    tau = T - t
    if model_type == "score_function":
        return -_b(path.f(tau), x) * x + 0.5 * _b(path.g2(tau), x) * fn(tau, x)
    if model_type == "noise_function":
        return -_b(path.f(tau), x) * x - 0.5 * _b(path.g2(tau), x) * fn(tau, x) / _b(path.std(tau), x)
    if model_type == "velocity_function":
        return -fn(tau, x)
    if model_type == "data_prediction_function":
        D = 0.5 * _b(path.g2(tau), x) / _b(path.cov(tau), x)
        return -(_b(path.f(tau), x) + D) * x + _b(path.scale(tau), x) * D * fn(tau, x)
    raise NotImplementedError(model_type)


def reverse_sde_drift(model_type, path, rpath, fn, t, x, T=1.0):
    """DiffusionProcess.reverse_sde drift, diffusion_process.py:742-851."""
    tau = T - t
    g2, gr2 = _b(path.g2(tau), x), _b(rpath.g2(tau), x)
    if model_type == "score_function":
        return -_b(path.f(tau), x) * x + 0.5 * (g2 + gr2) * fn(tau, x)
    if model_type == "noise_function":
        return -_b(path.f(tau), x) * x - 0.5 * (g2 + gr2) * fn(tau, x) / _b(path.std(tau), x)
    if model_type == "velocity_function":
        v = fn(tau, x)
        r = gr2 / (g2 + 1e-8)
        return -v - (v - _b(path.f(tau), x) * x) * r
    if model_type == "data_prediction_function":
        D = 0.5 * (g2 + gr2) / _b(path.cov(tau), x)
        return -(_b(path.f(tau), x) + D) * x + _b(path.scale(tau), x) * D * fn(tau, x)
    raise NotImplementedError(model_type)


def forward_ode_drift(model_type, path, fn, t, x):
    """DiffusionProcess.forward_ode, diffusion_process.py:672-740 (the model is evaluated at the solver time t)."""
    f, g2 = _b(path.f(t), x), _b(path.g2(t), x)
    if model_type == "score_function":
        return f * x - 0.5 * g2 * fn(t, x)
    if model_type == "noise_function":
        return f * x + 0.5 * g2 * fn(t, x) / _b(path.std(t), x)
    if model_type == "velocity_function":
        return fn(t, x)
    if model_type == "data_prediction_function":
        D = 0.5 * g2 / _b(path.cov(t), x)
        return (f + D) * x - _b(path.scale(t), x) * D * fn(t, x)
    raise NotImplementedError(model_type)


def forward_sde_drift(model_type, path, fpath, fn, t, x):
    """DiffusionProcess.forward_sde, diffusion_process.py:559-670; ``fpath`` is the process whose diffusion drives it."""
    f, g2, fg2 = _b(path.f(t), x), _b(path.g2(t), x), _b(fpath.g2(t), x)
    if model_type == "score_function":
        return f * x - 0.5 * (g2 - fg2) * fn(t, x)
    if model_type == "noise_function":
        return f * x + 0.5 * (g2 - fg2) * fn(t, x) / _b(path.std(t), x)
    if model_type == "velocity_function":
        v = fn(t, x)
        return v - (v - f * x) * (fg2 / (g2 + 1e-8))
    if model_type == "data_prediction_function":
        D = 0.5 * (g2 - fg2) / _b(path.cov(t), x)
        return (f + D) * x - _b(path.scale(t), x) * D * fn(t, x)
    raise NotImplementedError(model_type)


def forward_sample_process(sd, cfg, kind, x, t_span, condition=None, solver="ODESolver", dt=None, increments=None):
    """``forward_sample_process`` of DiffusionModel (diffusion_model.py:890-983) and of
    IndependentConditionalFlowModel (independent_conditional_flow_model.py:772-846): returns (T, N, D)."""
    fn = lambda t, y: intrinsic_forward(sd, t, y, condition)  # noqa: E731
    if kind == "icfm":
        lo, hi = t_span.min(), t_span.max()
        return euler_ode(lambda t, y: -fn(hi - t + lo, y), x, t_span)
    path, mt = Path(cfg["path"]), cfg["model"]["type"]
    if solver == "ODESolver":
        return euler_ode(lambda t, y: forward_ode_drift(mt, path, fn, t, y), x, t_span)
    return euler_maruyama(lambda t, y: forward_sde_drift(mt, path, path, fn, t, y),
                          lambda t, y: _b(path.g(t), y) if t.dim() else path.g(t) * torch.ones_like(y),
                          x, t_span, dt, increments)


def reverse_sde_diffusion(rpath, t, x, T=1.0):
    """diffusion_process.py:775-779 -- g_rev(T - t) broadcast to x."""
    return _b(rpath.g(T - t), x)


def energy_guidance(sd, t, x, cond, scale, pre="energy_guidance.model."):
    """EnergyGuidance.calculate_energy_guidance,
    diffusion_model/energy_conditional_diffusion_model.py:67-92: scale * d/dx sum_rows E(t,x,c)."""
    with torch.enable_grad():
        xg = x.detach().clone().requires_grad_(True)
        e = intrinsic_forward(sd, t, xg, cond, pre=pre, backbone="ConcatenateMLP")
        g = torch.autograd.grad(e.sum(), xg)[0]
    return (scale * g).detach()


def guided_score(sd, model_type, path, t, x, cond, scale):
    """score_function_with_energy_guidance, energy_conditional_diffusion_model.py:1109-1133."""
    net = intrinsic_forward(sd, t, x, cond)
    return score_head(model_type, path, net, t, x) + energy_guidance(sd, t, x, cond, scale)


# ----------------------------------------------------------------------------- integrators
def euler_ode(drift, x0, t_span):
    """Fixed-grid explicit Euler on exactly the caller's grid (torchdyn ``odeint(..,'euler')`` /
    torchdiffeq ``method='euler'`` as called from numerical_solvers/ode_solver.py:91-100,176-184);
    ``t`` reaches the drift repeated to (B,) (ode_solver.py:170-174).  Returns (T, B, D).
    STEPPING PARITY UNPINNED (third-party, see module docstring)."""
    traj = [x0]
    x = x0
    for k in range(len(t_span) - 1):
        tk = t_span[k].repeat(x.shape[0])
        x = x + (t_span[k + 1] - t_span[k]) * drift(tk, x)
        traj.append(x)
    return torch.stack(traj)


def euler_maruyama(drift, diffusion, x0, t_span, dt, increments, repeat_t=True):
    """torchsde ``sdeint(method='euler', dt=dt)`` as called from numerical_solvers/sde_solver.py:129-140:
    advance t <- min(t + dt, t_end) from t_span[0], diagonal Ito ``y += f dt + g dW``, outputs
    linearly interpolated at t_span; ``increments[k]`` is the k-th Brownian increment consumed.
    STEPPING PARITY UNPINNED."""
    dt = torch.as_tensor(dt, dtype=t_span.dtype)
    curr_t, curr_y = t_span[0], x0
    prev_t, prev_y = curr_t, curr_y
    ys = [x0]
    k = 0
    for out_t in t_span[1:]:
        while curr_t < out_t:
            next_t = torch.minimum(curr_t + dt, t_span[-1])
            prev_t, prev_y = curr_t, curr_y
            tt = curr_t.repeat(x0.shape[0]) if repeat_t else curr_t
            curr_y = curr_y + drift(tt, curr_y) * (next_t - curr_t) + diffusion(tt, curr_y) * increments[k]
            k += 1
            curr_t = next_t
        if curr_t == out_t or prev_t == curr_t:
            ys.append(curr_y)
        else:
            w = (out_t - prev_t) / (curr_t - prev_t)
            ys.append(prev_y + w * (curr_y - prev_y))
    return torch.stack(ys)


# ----------------------------------------------------------------------------- sampling front-end
def _expand_batch(batch_size, x_0, cond):
    """Batch bookkeeping of sample_forward_process, diffusion_model/diffusion_model.py:170-234."""
    if batch_size is None:
        extra = (1,)
    elif isinstance(batch_size, int):
        extra = (batch_size,)
    else:
        extra = tuple(int(b) for b in batch_size)
    if x_0 is not None:
        n = x_0.shape[0]
    elif cond is not None:
        n = cond.shape[0]
    else:
        n = 1
    return extra, n


def _reshape_out(data, batch_size, extra, n, x_size, x_0, cond):
    """diffusion_model.py:369-416: reshape (T, B*N, D) -> (T, *B, N, D) and squeeze (this reads
    rows laid out n*B+b as b*N+n -- reproduced as written)."""
    xs = (x_size,) if isinstance(x_size, int) else tuple(x_size)
    data = data.reshape(-1, *extra, n, *xs)
    if batch_size is None:
        data = data.squeeze(1).squeeze(1) if (x_0 is None and cond is None) else data.squeeze(1)
    elif x_0 is None and cond is None:
        data = data.squeeze(1 + len(extra))
    return data


def sample_forward_process(sd, cfg, kind, t_span, batch_size=None, x_0=None, condition=None,
                           x_T=None, guidance_scale=1.0, solver="ODESolver", dt=None, increments=None):
    """DiffusionModel / EnergyConditionalDiffusionModel / IndependentConditionalFlowModel
    ``sample_forward_process`` (diffusion_model.py:135-470, energy_conditional_diffusion_model.py:257-615,
    conditional_flow_model/independent_conditional_flow_model.py:170-450).  ``x_T`` replaces the
    draw from ``gaussian_generator`` (shape (prod(B)*N, D)) so runs are reproducible."""
    extra, n = _expand_batch(batch_size, x_0, condition)
    rep = int(math.prod(extra))
    x = x_T if x_0 is None else torch.repeat_interleave(x_0, rep, dim=0)
    cond = None if condition is None else torch.repeat_interleave(condition, rep, dim=0)
    mt = cfg["model"]["type"]
    if kind == "icfm":
        data = euler_ode(lambda t, y: intrinsic_forward(sd, t, y, cond), x, t_span)
    else:
        path = Path(cfg["path"])
        if kind == "energy":
            fn = lambda t, y: guided_score(sd, mt, path, t, y, cond, guidance_scale)  # noqa: E731
            ftype = "score_function"
        else:
            fn = lambda t, y: intrinsic_forward(sd, t, y, cond)  # noqa: E731
            ftype = mt
        if solver == "ODESolver":
            data = euler_ode(lambda t, y: reverse_ode_drift(ftype, path, fn, t, y), x, t_span)
        else:
            rpath = Path(cfg["reverse_path"])
            data = euler_maruyama(lambda t, y: reverse_sde_drift(ftype, path, rpath, fn, t, y),
                                  lambda t, y: reverse_sde_diffusion(rpath, t, y),
                                  x, t_span, dt, increments)
    return _reshape_out(data, batch_size, extra, n, cfg["x_size"], x_0, condition)


def sample_forward_process_with_fixed_x(sd, cfg, kind, t_span, fixed_x, fixed_mask, fixed_noise, batch_size=None,
                                        x_0=None, condition=None, x_T=None, guidance_scale=1.0, solver="ODESolver",
                                        dt=None, increments=None):
    """DiffusionModel / EnergyConditionalDiffusionModel ``sample_forward_process_with_fixed_x``
    (diffusion_model.py:521-859, energy_conditional_diffusion_model.py:669-1036): the drift (and the diffusion) are
    evaluated at ``direct_sample(t_max - t, fixed_x) * (1 - mask) + x * mask`` and multiplied by the mask
    (diffusion_model.py:647-661,725-752).  ``direct_sample`` (diffusion_process.py:939-956) draws fresh Gaussian noise
    at every call; ``fixed_noise[k]`` (n_steps, prod(B)*N, D) is the draw of the k-th DRIFT evaluation (the SDE
    branch's diffusion closure draws too, but g does not depend on x, so that draw only advances the generator)."""
    extra, _ = _expand_batch(batch_size, x_0, condition)
    n = fixed_x.shape[0]
    rep = int(math.prod(extra))
    x = x_T if x_0 is None else torch.repeat_interleave(x_0, rep, dim=0)
    cond = None if condition is None else torch.repeat_interleave(condition, rep, dim=0)
    fx = torch.repeat_interleave(fixed_x, rep, dim=0)
    m = torch.repeat_interleave(fixed_mask, rep, dim=0)
    mt = cfg["model"]["type"]
    path = Path(cfg["path"])
    if kind == "energy":
        fn = lambda t, y: guided_score(sd, mt, path, t, y, cond, guidance_scale)  # noqa: E731
        ftype = "score_function"
    else:
        fn = lambda t, y: intrinsic_forward(sd, t, y, cond)  # noqa: E731
        ftype = mt
    x = fx * (1 - m) + x * m
    calls = [0]

    def partially_fixed(t, y):
        tau = 1.0 - t
        eps = fixed_noise[calls[0]]
        calls[0] += 1
        return (_b(path.scale(tau), fx) * fx + _b(path.std(tau), fx) * eps) * (1 - m) + y * m

    if solver == "ODESolver":
        data = euler_ode(lambda t, y: m * reverse_ode_drift(ftype, path, fn, t, partially_fixed(t, y)), x, t_span)
    else:
        rpath = Path(cfg["reverse_path"])
        data = euler_maruyama(lambda t, y: m * reverse_sde_drift(ftype, path, rpath, fn, t, partially_fixed(t, y)),
                              lambda t, y: m * reverse_sde_diffusion(rpath, t, y),
                              x, t_span, dt, increments)
    xs = (cfg["x_size"],) if isinstance(cfg["x_size"], int) else tuple(cfg["x_size"])
    data = data.reshape(-1, *extra, n, *xs)
    return data.squeeze(1) if batch_size is None else data


# ----------------------------------------------------------------------------- losses
def _row_loss(pred, target, average):
    per_row = torch.sum(0.5 * (pred - target) ** 2, dim=1)
    return per_row.mean() if average else per_row


def score_matching_loss(model_type, path, net_fn, x, t, eps, average=True):
    """ScoreFunction.score_matching_loss, model_functions/score_function.py:192-251, with the
    random draws (t first, then eps) passed in explicitly."""
    std, scale = _b(path.std(t), x), _b(path.scale(t), x)
    x_t = scale * x + std * eps
    net = net_fn(t, x_t)
    if model_type == "noise_function":
        pred = net
    elif model_type == "score_function":
        pred = net * std
    elif model_type == "velocity_function":
        pred = (net - _b(path.f(t), x) * x_t) * 2.0 * std / _b(path.g2(t), x)
    elif model_type == "data_prediction_function":
        pred = (x_t - scale * net) / std
    else:
        raise NotImplementedError(model_type)
    return _row_loss(pred, eps, average)


def flow_matching_loss(model_type, path, net_fn, x, t, eps, average=True):
    """VelocityFunction.flow_matching_loss, model_functions/velocity_function.py:131-214."""
    std, scale = _b(path.std(t), x), _b(path.scale(t), x)
    x_t = scale * x + std * eps
    net = net_fn(t, x_t)
    fx = _b(path.f(t), x) * x_t
    g2 = _b(path.g2(t), x)
    if model_type == "noise_function":
        pred = fx + 0.5 * g2 * net / std
    elif model_type == "score_function":
        pred = fx - 0.5 * g2 * net
    elif model_type == "velocity_function":
        pred = net
    elif model_type == "data_prediction_function":
        D = 0.5 * g2 / _b(path.cov(t), x)
        pred = (_b(path.f(t), x) + D) * x_t - D * scale * net
    else:
        raise NotImplementedError(model_type)
    target = _b(path.d_scale(t), x) * x + _b(path.d_std(t), x) * eps  # diffusion_process.py:405-417
    return _row_loss(pred, target, average)


def icfm_flow_matching_loss(net_fn, x0, x1, t, eps, sigma, average=True):
    """VelocityFunction.flow_matching_loss_icfm, velocity_function.py:289-296 with
    StochasticProcess.direct_sample / velocity, generative_models/stochastic_process.py:26-129."""
    tb = _b(t, x0)
    x_t = x0 * (1.0 - tb) + x1 * tb + sigma * eps
    return _row_loss(net_fn(t, x_t), x1 - x0, average)


def icfm_flow_matching_loss_with_mask(net_fn, x0, x1, t, eps, sigma, mask, average=True):
    """IndependentConditionalFlowModel.flow_matching_loss_with_mask, independent_conditional_flow_model.py:699-741:
    element-wise loss (velocity_function.py:248-253), elements outside the mask dropped, per-row mean over the kept
    elements (at least 1)."""
    tb = _b(t, x0)
    x_t = x0 * (1.0 - tb) + x1 * tb + sigma * eps
    v = net_fn(t, x_t)
    el = 0.5 * (v - (x1 - x0)) * (v - (x1 - x0))
    loss = el.masked_fill(~mask, 0.0).sum(1) / torch.clamp(mask.sum(1), min=1)
    return (loss.mean() if average else loss), el


def log_likelihood(sd, cfg, kind, x, t_span, condition=None, hutchinson=True, probes=None):
    """compute_likelihood (grl/generative_models/metric.py:14-139) / IndependentConditionalFlowModel.log_prob
    (independent_conditional_flow_model.py:515-631): Euler on t_span of d(x, logp)/dt = (f, -div f) with
    f = forward_ode drift (diffusion) or -model(1 - t, x, c) (flow), div f exact (one autograd product per dimension,
    metric.py:56-76) or Hutchinson-Skilling with the Gaussian probes ``probes[k]`` (metric.py:78-84), then the standard
    normal log-density at the end point minus the accumulated difference."""
    fn = lambda t, y: intrinsic_forward(sd, t, y, condition)  # noqa: E731
    if kind == "icfm":
        drift = lambda t, y: -fn(1.0 - t, y)  # noqa: E731
    else:
        path, mt = Path(cfg["path"]), cfg["model"]["type"]
        drift = lambda t, y: forward_ode_drift(mt, path, fn, t, y)  # noqa: E731
    x = x.clone()
    diff = torch.zeros(x.shape[0])
    for k in range(len(t_span) - 1):
        tk = t_span[k].repeat(x.shape[0])
        with torch.enable_grad():
            xt = x.detach().requires_grad_(True)
            dx = drift(tk, xt)
            if hutchinson:
                eps = probes[k]
                g = torch.autograd.grad(torch.sum(dx * eps), xt)[0]
                div = torch.sum(g * eps, dim=1)
            else:
                div = torch.zeros(x.shape[0])
                for i in range(x.shape[1]):
                    go = torch.zeros_like(xt)
                    go[:, i] = 1
                    div = div + torch.autograd.grad(dx, xt, grad_outputs=go, retain_graph=True)[0][:, i]
        dt = t_span[k + 1] - t_span[k]
        diff = diff + dt * (-div.detach())
        x = x + dt * dx.detach()
    logp_end = -0.5 * (x * x).sum(1) - 0.5 * x.shape[1] * math.log(2.0 * math.pi)
    return logp_end - diff


# ----------------------------------------------------------------------------- DPM-Solver
def dpm_orders_singlestep(steps, order):
    """get_orders_for_singlestep_solver, numerical_solvers/dpm_solver.py:163-230."""
    if order == 3:
        K = steps // 3 + 1
        if steps % 3 == 0:
            return [3] * (K - 2) + [2, 1]
        if steps % 3 == 1:
            return [3] * (K - 1) + [1]
        return [3] * (K - 1) + [2]
    if order == 2:
        if steps % 2 == 0:
            K = steps // 2 + 1
            return [2] * (K - 2) + [1] * 2
        return [2] * (steps // 2) + [1]
    if order == 1:
        return [1] * steps
    raise ValueError(order)


def dpm_time_steps(path, t_T, t_0, N, skip_type="time_uniform"):
    """get_time_steps, dpm_solver.py:127-161."""
    if skip_type == "logSNR":
        lT, l0 = path.half_log_snr(torch.tensor(t_T)), path.half_log_snr(torch.tensor(t_0))
        return path.inverse_half_log_snr(torch.linspace(float(lT), float(l0), N + 1))
    if skip_type == "time_uniform":
        return torch.linspace(t_T, t_0, N + 1)
    if skip_type == "time_quadratic":
        return torch.linspace(t_T ** 0.5, t_0 ** 0.5, N + 1).pow(2)
    raise ValueError(skip_type)


def dpm_solver_integrate(path, noise_fn, data_fn, x, steps=20, order=2, type="dpm_solver", method="singlestep",
                         solver_type="dpm_solver", skip_type="time_uniform", denoise=False):
    """DPMSolver.integrate with method 'singlestep' / 'singlestep_fixed' (dpm_solver.py:83-905: first update
    :250-299, second :301-391, third :393-523, driver :854-904).  ``noise_fn(t, x)`` / ``data_fn(t, x)`` take
    t of shape (B,).  Returns the stacked states (len(orders) + 1, B, ...) like ``save_intermediate=True``."""
    plus = type == "dpm_solver++"
    high = solver_type == "dpm_solver"
    B = x.shape[0]
    model_fn = (lambda xx, tt: data_fn(tt.expand(B), xx)) if plus else (lambda xx, tt: noise_fn(tt.expand(B), xx))
    lam, la, sig = path.half_log_snr, path.log_scale, path.std

    def first(x, s, t):
        h = lam(t) - lam(s)
        m = model_fn(x, s)
        if plus:
            return sig(t) / sig(s) * x - torch.exp(la(t)) * torch.expm1(-h) * m
        return torch.exp(la(t) - la(s)) * x - sig(t) * torch.expm1(h) * m

    def second(x, s, t, r1):
        r1 = 0.5 if r1 is None else r1
        h = lam(t) - lam(s)
        s1 = path.inverse_half_log_snr(lam(s) + r1 * h)[0]
        m = model_fn(x, s)
        if plus:
            p11, p1 = torch.expm1(-r1 * h), torch.expm1(-h)
            x1 = sig(s1) / sig(s) * x - torch.exp(la(s1)) * p11 * m
            m1 = model_fn(x1, s1)
            at = torch.exp(la(t))
            if high:
                return sig(t) / sig(s) * x - at * p1 * m - (0.5 / r1) * at * p1 * (m1 - m)
            return sig(t) / sig(s) * x - at * p1 * m + (1.0 / r1) * at * ((torch.exp(-h) - 1.0) / h + 1.0) * (m1 - m)
        p11, p1 = torch.expm1(r1 * h), torch.expm1(h)
        x1 = torch.exp(la(s1) - la(s)) * x - sig(s1) * p11 * m
        m1 = model_fn(x1, s1)
        e = torch.exp(la(t) - la(s))
        if high:
            return e * x - sig(t) * p1 * m - (0.5 / r1) * sig(t) * p1 * (m1 - m)
        return e * x - sig(t) * p1 * m - (1.0 / r1) * sig(t) * ((torch.exp(h) - 1.0) / h - 1.0) * (m1 - m)

    def third(x, s, t, r1, r2):
        r1 = 1.0 / 3.0 if r1 is None else r1
        r2 = 2.0 / 3.0 if r2 is None else r2
        h = lam(t) - lam(s)
        s1 = path.inverse_half_log_snr(lam(s) + r1 * h)[0]
        s2 = path.inverse_half_log_snr(lam(s) + r2 * h)[0]
        m = model_fn(x, s)
        if plus:
            p11, p12, p1 = torch.expm1(-r1 * h), torch.expm1(-r2 * h), torch.expm1(-h)
            p22 = torch.expm1(-r2 * h) / (r2 * h) + 1.0
            p2 = p1 / h + 1.0
            p3 = p2 / h - 0.5
            x1 = sig(s1) / sig(s) * x - torch.exp(la(s1)) * p11 * m
            m1 = model_fn(x1, s1)
            a2, at = torch.exp(la(s2)), torch.exp(la(t))
            x2 = sig(s2) / sig(s) * x - a2 * p12 * m + r2 / r1 * a2 * p22 * (m1 - m)
            m2 = model_fn(x2, s2)
            if high:
                return sig(t) / sig(s) * x - at * p1 * m + (1.0 / r2) * at * p2 * (m2 - m)
            D10, D11 = (1.0 / r1) * (m1 - m), (1.0 / r2) * (m2 - m)
            D1, D2 = (r2 * D10 - r1 * D11) / (r2 - r1), 2.0 * (D11 - D10) / (r2 - r1)
            return sig(t) / sig(s) * x - at * p1 * m + at * p2 * D1 - at * p3 * D2
        p11, p12, p1 = torch.expm1(r1 * h), torch.expm1(r2 * h), torch.expm1(h)
        p22 = torch.expm1(r2 * h) / (r2 * h) - 1.0
        p2 = p1 / h - 1.0
        p3 = p2 / h - 0.5
        x1 = torch.exp(la(s1) - la(s)) * x - sig(s1) * p11 * m
        m1 = model_fn(x1, s1)
        x2 = torch.exp(la(s2) - la(s)) * x - sig(s2) * p12 * m - r2 / r1 * sig(s2) * p22 * (m1 - m)
        m2 = model_fn(x2, s2)
        e = torch.exp(la(t) - la(s))
        if high:
            return e * x - sig(t) * p1 * m - (1.0 / r2) * sig(t) * p2 * (m2 - m)
        D10, D11 = (1.0 / r1) * (m1 - m), (1.0 / r2) * (m2 - m)
        D1, D2 = (r2 * D10 - r1 * D11) / (r2 - r1), 2.0 * (D11 - D10) / (r2 - r1)
        return e * x - sig(t) * p1 * m - sig(t) * p2 * D1 - sig(t) * p3 * D2

    t_0, t_T = 0.00001, path.t_max
    if method == "singlestep":
        orders = dpm_orders_singlestep(steps, order)
        ts = dpm_time_steps(path, t_T, t_0, steps, skip_type)
    elif method == "singlestep_fixed":
        K = steps // order
        orders = [order] * K
        ts = dpm_time_steps(path, t_T, t_0, K * order, skip_type)
    else:
        raise NotImplementedError(method)
    xs = [x.clone()]
    i = 0
    for o in orders:
        s, t = ts[i], ts[i + o]
        h = lam(ts[i + o]) - lam(ts[i])
        r1 = None if o <= 1 else (lam(ts[i + 1]) - lam(ts[i])) / h
        r2 = None if o <= 2 else (lam(ts[i + 2]) - lam(ts[i])) / h
        x = first(x, s, t) if o == 1 else second(x, s, t, r1) if o == 2 else third(x, s, t, r1, r2)
        xs.append(x.clone())
        i += o
    if denoise:
        x = data_fn(torch.ones(B) * t_0, x)
        xs[-1] = x.clone()
    return torch.stack(xs, 0)


# ----------------------------------------------------------------------------- CEP
def energy_guidance_loss(sd, path, x, cond, energy, t, noise, alpha=1.0, pre="energy_guidance.model."):
    """EnergyConditionalDiffusionModel.energy_guidance_loss,
    diffusion_model/energy_conditional_diffusion_model.py:1308-1394 (contrastive energy prediction).
    x (B, K, D) support actions, cond (B, C) | None, energy (B, K) = energy_model(x, cond) detached,
    t (B,) and noise (B, K, D) = the reference's draws (:1326 and diffusion_process.py:956).
    sd values that require grad receive the gradient when the result is back-propagated."""
    B, K, D = x.shape
    t_rep = torch.stack([t] * K, dim=1)                                               # :1327
    x_t = path.scale(t_rep)[..., None] * x + path.std(t_rep)[..., None] * noise       # :1354
    c_rep = None if cond is None else torch.stack([cond] * K, dim=1).reshape(B * K, -1)  # :1368-1372
    f = intrinsic_forward(sd, t_rep.reshape(B * K), x_t.reshape(B * K, D), c_rep, pre=pre,
                          backbone="ConcatenateMLP").reshape(B, K)                    # :1375-1380
    log_p = torch.log_softmax(f, dim=1)                                               # :1389
    p0 = torch.softmax(energy * alpha, dim=1)                                         # :1390
    return -torch.mean(torch.sum(p0 * log_p, dim=-1))                                 # :1391-1393


def q_network_forward(sd, pre, action, state):
    """QNetwork.forward, grl/rl_modules/value_network/q_network.py:48-64: ConcatenateMLP(ReLU) over cat[action, state]
    (MultiLayerPerceptron, grl/neural_network/__init__.py:475-555)."""
    h = torch.cat([action, state], dim=-1)
    i = 0
    while (pre + "model.backbone.model.model.linear%d.weight" % (i + 1)) in sd:
        h = F.relu(_lin(sd, pre + "model.backbone.model.model.linear%d" % i, h))
        i += 1
    return _lin(sd, pre + "model.backbone.model.model.linear%d" % i, h)


def q_loss(sd, action, state, reward, next_state, done, fake_next_action, q_alpha=1.0, discount_factor=1.0):
    """QGPOCritic.q_loss, grl/algorithms/qgpo.py:85-139.  sd: the critic's state_dict ("q.model.q1...." and
    "q_target.model.q1...."); values that require grad receive the gradient on backward.
    PARITY UNPINNED for this composition (grl/algorithms/qgpo.py cannot be imported here: torchrl, gym, d4rl are
    absent); the networks it calls are pinned by tests/golden/qloss.npz, made with the reference's DoubleQNetwork."""
    with torch.no_grad():
        K = fake_next_action.shape[1]
        ns = torch.stack([next_state] * K, dim=1)                                         # :118-120
        e = torch.min(q_network_forward(sd, "q_target.model.q1.", fake_next_action, ns),
                      q_network_forward(sd, "q_target.model.q2.", fake_next_action, ns)).squeeze(dim=-1)  # :121-128
        next_v = torch.sum(torch.softmax(q_alpha * e, dim=1) * e, dim=-1, keepdim=True)   # :129-131
    targets = reward + (1.0 - done.float()) * discount_factor * next_v                    # :133
    q0 = q_network_forward(sd, "q.model.q1.", action, state)
    q1 = q_network_forward(sd, "q.model.q2.", action, state)
    return (F.mse_loss(q0, targets) + F.mse_loss(q1, targets)) / 2                        # :135-138


# ----------------------------------------------------------------------------- GMPO / IQL critic
def v_network_forward(sd, pre, state):
    """VNetwork.forward, grl/rl_modules/value_network/value_network.py:48-70 with a MultiLayerPerceptron(ReLU)
    backbone (grl/neural_network/__init__.py:475-555)."""
    h = state
    i = 0
    while (pre + "model.backbone.model.linear%d.weight" % (i + 1)) in sd:
        h = F.relu(_lin(sd, pre + "model.backbone.model.linear%d" % i, h))
        i += 1
    return _lin(sd, pre + "model.backbone.model.linear%d" % i, h)


def asymmetric_l2_loss(u, tau):
    """grl/algorithms/gmpo.py:41-42"""
    return torch.mean(torch.abs(tau - (u < 0).float()) * u ** 2)


def _min_q(sd, pre, action, state):
    return torch.min(q_network_forward(sd, pre + "model.q1.", action, state), q_network_forward(sd, pre + "model.q2.", action, state))


def gmpo_v_loss(sd, state, action, next_state, tau):
    """GMPOCritic.v_loss, grl/algorithms/gmpo.py:100-108 -> (v_loss, next_v).  PARITY UNPINNED for the composition
    (gmpo.py cannot be imported here); networks pinned by tests/golden/iql.npz (reference DoubleQNetwork / VNetwork)."""
    with torch.no_grad():
        target_q = _min_q(sd, "q_target.", action, state)
        next_v = v_network_forward(sd, "v.", next_state)
    adv = target_q - v_network_forward(sd, "v.", state)
    return asymmetric_l2_loss(adv, tau), next_v


def gmpo_iql_q_loss(sd, state, action, reward, done, next_v, discount):
    """GMPOCritic.iql_q_loss, grl/algorithms/gmpo.py:110-114 -> (q_loss, mean q1, mean target).  PARITY UNPINNED for
    the composition, as gmpo_v_loss."""
    q_target = reward + (1.0 - done.float()) * discount * next_v.detach()
    qs = (q_network_forward(sd, "q.model.q1.", action, state), q_network_forward(sd, "q.model.q2.", action, state))
    return sum(F.mse_loss(q, q_target) for q in qs) / len(qs), torch.mean(qs[0]), torch.mean(q_target)


def awr_weights(sd, action, state, beta=1.0, weight_clamp=100.0):
    """the weights of GMPOPolicy.policy_optimization_loss_by_advantage_weighted_regression,
    grl/algorithms/gmpo.py:323-338 -> (clamped_weight, weight); critic keys "critic.q....", "critic.v....".
    PARITY UNPINNED (no reference output available for it here)."""
    with torch.no_grad():
        q_value = _min_q(sd, "critic.q.", action, state).squeeze(dim=-1)
        v_value = v_network_forward(sd, "critic.v.", state).squeeze(dim=-1)
        weight = torch.exp(beta * (q_value - v_value))
    return weight.clamp(max=weight_clamp), weight
