"""The training losses of the hot path, reduced to ONE fused forward+backward kernel call.

Every variant of ScoreFunction.score_matching_loss (grl/generative_models/model_functions/
score_function.py:192-251), VelocityFunction.flow_matching_loss (velocity_function.py:131-214) and
flow_matching_loss_icfm (:289-296) is
    pred = alpha(t) * net(t, x_t, c) + beta(t) * x_t ;   loss_row = sum_d 0.5 * (pred - target)^2
The ``prepare_*`` functions draw t / noise in the reference's order (t first, then the noise),
build x_t, alpha, beta and target with a handful of elementwise device ops, and ``fused_loss`` runs
the network forward+backward in the HIP library (``grlhip_loss_fwd_bwd``), exposing the result to
autograd so ``loss.backward()`` fills ``.grad`` of the ordinary parameters."""
import ctypes
import weakref

import torch

from ... import _lib as L
from ...engine import _f32c, _ptr, _stream
from ...numerical_methods.monte_carlo import MonteCarloSampler


def _bc(v, x):
    return v[(...,) + (None,) * (x.dim() - v.dim())].expand(x.shape) if x.dim() > v.dim() else v


def _draw_t(fn_owner, process, batch_size, device, weighting_scheme):
    if weighting_scheme is None or weighting_scheme == "maximum_likelihood":
        if not hasattr(fn_owner, "monte_carlo_sampler"):  # score_function.py:129-143
            fn_owner.monte_carlo_sampler = MonteCarloSampler(
                unnormalized_pdf=lambda t: process.path.diffusion_squared(t) / (process.path.covariance(t) + 1e-6),
                x_min=0.0, x_max=process.t_max, device=device)
        return fn_owner.monte_carlo_sampler.sample(batch_size).to(device=device)
    if weighting_scheme == "vanilla":
        eps = 1e-5
        return torch.rand(batch_size, device=device) * (process.t_max - eps) + eps
    raise NotImplementedError("Unknown weighting scheme {}".format(weighting_scheme))


_CONST = {}


def _const(value, n, dev):
    """A constant fp32 vector of n elements (a unit alpha, a zero beta, the 1 / B row weights): read-only operands of
    the library calls, built once per (value, n, device) instead of one fill launch per training step."""
    key = (float(value), int(n), str(dev))
    v = _CONST.get(key)
    if v is None:
        v = torch.full((int(n),), float(value), dtype=torch.float32, device=dev)
        # a fill recorded into a HIP graph under capture has not run yet: such a tensor is not kept for eager callers
        if not (v.is_cuda and torch.cuda.is_current_stream_capturing()):
            if len(_CONST) >= 64:
                _CONST.clear()
            _CONST[key] = v
    return v


def prepare_score_matching(owner, model_type, process, x, gaussian_generator, weighting_scheme, t=None, noise=None):
    B, dev = x.shape[0], x.device
    if t is None:
        t = _draw_t(owner, process, B, dev, weighting_scheme)
    if noise is None:
        noise = torch.randn_like(x) if gaussian_generator is None else gaussian_generator(B)
    p = process.path
    std, scale = p.std(t), p.scale(t)
    x_t = _bc(scale, x) * x + _bc(std, x) * noise
    one, zero = _const(1.0, B, dev), _const(0.0, B, dev)
    if model_type == "noise_function":
        alpha, beta = one, zero
    elif model_type == "score_function":
        alpha, beta = std, zero
    elif model_type == "velocity_function":
        k = 2.0 * std / p.diffusion_squared(t)
        alpha, beta = k, -p.drift_coefficient(t) * k
    elif model_type == "data_prediction_function":
        alpha, beta = -scale / std, one / std
    else:
        raise NotImplementedError("Unknown type of score function {}".format(model_type))
    return t, x_t, alpha, beta, noise


def prepare_flow_matching(model_type, process, x, gaussian_generator, t=None, noise=None):
    B, dev = x.shape[0], x.device
    eps = 1e-5
    if t is None:
        t = torch.rand(B, device=dev) * (process.t_max - eps) + eps
    if noise is None:
        noise = torch.randn_like(x) if gaussian_generator is None else gaussian_generator(B)
    p = process.path
    std, scale = p.std(t), p.scale(t)
    x_t = _bc(scale, x) * x + _bc(std, x) * noise
    f, g2 = p.drift_coefficient(t), p.diffusion_squared(t)
    one, zero = _const(1.0, B, dev), _const(0.0, B, dev)
    if model_type == "noise_function":
        alpha, beta = 0.5 * g2 / std, f
    elif model_type == "score_function":
        alpha, beta = -0.5 * g2, f
    elif model_type == "velocity_function":
        alpha, beta = one, zero
    elif model_type == "data_prediction_function":
        D = 0.5 * g2 / p.covariance(t)
        alpha, beta = -D * scale, f + D
    else:
        raise NotImplementedError("Unknown type of velocity function {}".format(model_type))
    target = process.velocity(t, x, noise=noise)
    return t, x_t, alpha, beta, target


def prepare_icfm(process, x0, x1, t=None, noise=None):
    """flow_matching_loss_icfm (velocity_function.py:289-296) + StochasticProcess.direct_sample
    (stochastic_process.py:108-129): t = rand(B) * t_max, x_t = x0 (1-t) + x1 t + sigma * randn."""
    B, dev = x0.shape[0], x0.device
    if t is None:
        t = torch.rand(B, device=dev)
        if float(process.t_max) != 1.0:
            t = t * process.t_max
    if noise is None:
        noise = torch.randn_like(x0)
    sigma = getattr(getattr(process.path, "config", None), "sigma", None)
    if sigma is not None and x0.dim() == 2 and t.dim() == 1:
        # the same x_t in two launches instead of eight (interpolation, then + sigma * noise); fp32 rounding level
        x_t = torch.add(torch.lerp(x0, x1, t.unsqueeze(1)), noise, alpha=float(sigma))
    else:
        x_t = process.mean(t, x0, x1) + process.std(t, x0, x1) * noise
    return t, x_t, _const(1.0, B, dev), _const(0.0, B, dev), process.velocity(t, x0, x1)


def _param_grads(model, flat, n_params):
    """Views of the library's flat gradient, one per parameter in plan order (None where no grad is required)."""
    plan = model.plan
    named = model._named()
    grads = []
    for i, (name, shape) in enumerate(plan.params):
        p = named[name]
        if p.requires_grad and flat is not None:
            off = L.lib().grlhip_grad_offset(plan.handle, i)
            grads.append(flat[off:off + p.numel()].view(p.shape))
        else:
            grads.append(None)
    assert len(grads) == n_params
    return grads


class _StageToken:
    """Lives as long as the autograd node whose backward still needs the staged gradient."""
    __slots__ = ("__weakref__", "gen")


def _stage(model, n, dev):
    """Zeroed fp32 buffer for this call's own gradient contribution + the token that keeps it reserved.  The
    buffer is cached on the model and reused once the previous call's autograd node is gone (its backward ran or its
    graph was dropped); while that node is alive -- two losses of one model summed before one backward -- a fresh
    buffer is handed out instead."""
    ref = getattr(model, "_stage_token", None)
    if ref is not None and ref() is not None:
        return torch.zeros(n, dtype=torch.float32, device=dev), None
    st = getattr(model, "_grad_stage", None)
    if st is None or st.numel() != n or st.device != torch.device(dev):
        st = torch.empty(n, dtype=torch.float32, device=dev)
        model._grad_stage = st
    st.zero_()
    token = _StageToken()
    model._stage_token = weakref.ref(token)
    model._stage_gen = getattr(model, "_stage_gen", 0) + 1
    token.gen = model._stage_gen
    return st, token


def _consume(ctx):
    """Backward has read the staged gradient: give the cached buffer back (in the usual loop the loss tensor of
    step i is still bound while step i + 1 runs its forward, which would otherwise get a fresh buffer every second
    step).  A second backward through the same node (retain_graph=True) is still fine as long as no other loss of
    this model ran in between; after one did, the buffer holds that call's gradient and this raises."""
    token = ctx.token
    if token is None:
        if getattr(ctx, "stage_gen", None) is not None and ctx.flat is getattr(ctx.model, "_grad_stage", None) \
                and ctx.stage_gen != getattr(ctx.model, "_stage_gen", None):
            raise RuntimeError("fused loss: backward called again (retain_graph) after another loss of the same model "
                               "reused the staged gradient buffer; not supported")
        return
    ctx.stage_gen = token.gen
    ctx.token = None


class _FusedLoss(torch.autograd.Function):
    """forward = ONE grlhip_loss_fwd_bwd call (loss and, when a parameter requires grad under an enabled grad mode,
    the gradient of the mean loss into a staging buffer); backward scales the staged gradient by grad_output.
    With a FlatGrads buffer attached to the model (distributed.attach_flat_grads) the backward is one fused
    ``flat += g * staged`` pass into the buffer the parameters' ``.grad`` alias, and autograd receives no
    per-parameter tensors at all."""

    @staticmethod
    def forward(ctx, model, t, x_t, cond, alpha, beta, target, average, grad_mode, *params):
        plan = model.plan
        B = x_t.shape[0]
        dev = x_t.device
        lib = L.lib()
        loss_rows = torch.empty(B, dtype=torch.float32, device=dev)
        n_grad = lib.grlhip_grad_numel(plan.handle)
        need_grad = grad_mode and any(p.requires_grad for p in params)
        flat, ctx.token = _stage(model, n_grad, dev) if (need_grad and average) else (None, None)
        roww = _const(1.0 / B, B, dev) if flat is not None else None
        _launch(model, t, x_t, cond, alpha, beta, target, roww, loss_rows, flat)
        ctx.model, ctx.average, ctx.flat = model, average, flat
        ctx.n_params = len(params)
        if not average:
            ctx.save_for_backward(t, x_t, cond if cond is not None else torch.empty(0, device=dev), alpha, beta, target)
            ctx.has_cond = cond is not None
            return loss_rows
        return loss_rows.mean()

    @staticmethod
    def backward(ctx, gout):
        model = ctx.model
        plan = model.plan
        if ctx.average:
            flat = ctx.flat
            scale = gout
            _consume(ctx)
        else:
            t, x_t, cond, alpha, beta, target = ctx.saved_tensors
            flat = torch.zeros(L.lib().grlhip_grad_numel(plan.handle), dtype=torch.float32, device=x_t.device)
            scratch = torch.empty(x_t.shape[0], dtype=torch.float32, device=x_t.device)
            _launch(model, t, x_t, cond if ctx.has_cond else None, alpha, beta, target,
                    gout.contiguous().float(), scratch, flat)
            scale = None
        none8 = (None,) * 9
        fg = getattr(model, "_flat_grads", None)
        if fg is not None and flat is not None:
            fg.ensure()
            if scale is None:
                fg.grads.add_(flat)
            else:
                torch.addcmul(fg.grads, flat, scale.reshape(1).expand_as(flat), out=fg.grads)
            return (*none8, *([None] * ctx.n_params))
        if flat is not None:
            flat = flat * scale if scale is not None else flat.clone()
        return (*none8, *_param_grads(model, flat, ctx.n_params))


def _launch(model, t, x_t, cond, alpha, beta, target, roww, loss_rows, flat):
    plan = model.plan
    B, dev = x_t.shape[0], x_t.device
    lib = L.lib()
    t, x_t, cond = _f32c(t, "t", (B,)), _f32c(x_t, "x_t", (B, plan.x_dim)), _f32c(cond, "condition")
    alpha, beta, target = _f32c(alpha, "alpha", (B,)), _f32c(beta, "beta", (B,)), _f32c(target, "target", (B, plan.out_dim))
    ws = plan.train_workspace(B, dev)
    with torch.cuda.device(dev):
        L.check(lib.grlhip_loss_fwd_bwd(plan.handle, _ptr(model.packed(L.F32)), L.F32, _ptr(t), _ptr(x_t), _ptr(cond),
                                        _ptr(alpha), _ptr(beta), _ptr(target), _ptr(roww), _ptr(loss_rows),
                                        _ptr(flat), B, _ptr(ws), ws.numel(), _stream(dev)))


def fused_loss(model, t, x_t, cond, alpha, beta, target, average=True):
    """model: IntrinsicModel.  Returns the scalar mean loss (average=True) or the per-row losses."""
    named = model._named()
    params = [named[n] for n, _ in model.plan.params]
    return _FusedLoss.apply(model, t, x_t, cond, alpha, beta, target, average, torch.is_grad_enabled(), *params)


class _FusedCepLoss(torch.autograd.Function):
    """Contrastive energy prediction: energy-guidance net forward + group softmax cross-entropy + backward in ONE
    library call (grlhip_cep_loss_fwd_bwd); the mean over groups is returned."""

    @staticmethod
    def forward(ctx, model, t, x_t, cond, logits0, group_size, grad_mode, *params):
        plan = model.plan
        R, dev = x_t.shape[0], x_t.device
        G = R // group_size
        lib = L.lib()
        loss_groups = torch.empty(G, dtype=torch.float32, device=dev)
        need_grad = grad_mode and any(p.requires_grad for p in params)
        flat, ctx.token = _stage(model, lib.grlhip_grad_numel(plan.handle), dev) if need_grad else (None, None)
        gw = _const(1.0 / G, G, dev) if need_grad else None
        t, x_t, cond = _f32c(t, "t", (R,)), _f32c(x_t, "x_t", (R, plan.x_dim)), _f32c(cond, "condition")
        logits0 = _f32c(logits0, "logits0", (R,))
        ws = plan.train_workspace(R, dev)
        with torch.cuda.device(dev):
            L.check(lib.grlhip_cep_loss_fwd_bwd(plan.handle, _ptr(model.packed(L.F32)), L.F32, _ptr(t), _ptr(x_t),
                                                _ptr(cond), _ptr(logits0), group_size, _ptr(gw), _ptr(loss_groups),
                                                _ptr(flat), R, _ptr(ws), ws.numel(), _stream(dev)))
        ctx.model, ctx.flat, ctx.n_params = model, flat, len(params)
        return loss_groups.mean()

    @staticmethod
    def backward(ctx, gout):
        model, flat = ctx.model, ctx.flat
        _consume(ctx)
        fg = getattr(model, "_flat_grads", None)
        if fg is not None and flat is not None:
            fg.ensure()
            torch.addcmul(fg.grads, flat, gout.reshape(1).expand_as(flat), out=fg.grads)
            return (*((None,) * 7), *([None] * ctx.n_params))
        return (*((None,) * 7), *_param_grads(model, None if flat is None else flat * gout, ctx.n_params))


def fused_cep_loss(model, t, x_t, cond, logits0, group_size):
    """model: the energy guidance's IntrinsicModel (ConcatenateMLP).  Rows are B groups of ``group_size``
    consecutive support actions; returns -mean_b sum_k softmax(logits0)_bk log_softmax(f(t, x_t, c))_bk."""
    named = model._named()
    params = [named[n] for n, _ in model.plan.params]
    return _FusedCepLoss.apply(model, t, x_t, cond, logits0, group_size, torch.is_grad_enabled(), *params)
