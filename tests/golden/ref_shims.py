"""Stand-ins that let the reference's own classes import in this container: the third-party
packages the image lacks are replaced by minimal modules that are only reached on untaken branches, and the
fixed-grid Euler / Euler-Maruyama integrators are restated from the documented behaviour of torchdyn /
torchdiffeq / torchsde.  Importing this module installs them and puts /root/reference on sys.path.
Used by make_golden.py and make_golden_dpm.py (container-only scripts; nothing under tests/ imports it)."""
import sys
import types

import torch

# --------------------------------------------------------------------------- stand-ins
class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, cls):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(i) for i in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__


def _install_standins():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("easydict", EasyDict=EasyDict)
    mod("tensordict", TensorDict=type("TensorDict", (), {}))
    tt_torch = mod("treetensor.torch", Tensor=type("Tensor", (), {}), Size=type("Size", (), {}))
    mod("treetensor", torch=tt_torch)
    mod("ot")
    mod("wandb")
    mod("grl.neural_network.transformers.maxvit", MaxVit=type("MaxVit", (torch.nn.Module,), {}),
        MaxViT_t=type("MaxViT_t", (torch.nn.Module,), {}))

    # ---- integrators (specification, see module docstring) ----
    def euler_fixed_grid(f, x, t_span):
        if isinstance(x, tuple):   # torchdiffeq integrates tuples of tensors component-wise and returns a tuple
            traj = [x]
            for k in range(len(t_span) - 1):
                dx = f(t_span[k], x)
                x = tuple(xi + (t_span[k + 1] - t_span[k]) * di for xi, di in zip(x, dx))
                traj.append(x)
            return tuple(torch.stack([s[i] for s in traj]) for i in range(len(x)))
        traj = [x]
        for k in range(len(t_span) - 1):
            x = x + (t_span[k + 1] - t_span[k]) * f(t_span[k], x)
            traj.append(x)
        return torch.stack(traj)

    def torchdyn_odeint(f, x, t_span, solver, atol=1e-5, rtol=1e-5, **kw):
        assert solver == "euler"
        return t_span, euler_fixed_grid(f, x, t_span)

    def torchdiffeq_odeint(func, y0, t, method=None, atol=None, rtol=None, **kw):
        assert method == "euler"
        return euler_fixed_grid(func, y0, t)

    def torchsde_sdeint(sde, y0, ts, method="euler", dt=1e-3, bm=None, **kw):
        """dt-driven Euler-Maruyama from ts[0], clipped at ts[-1], outputs linearly
        interpolated at ts (torchsde BaseSDESolver.integrate semantics, diagonal Ito)."""
        assert method == "euler" and bm is not None
        dt = torch.as_tensor(dt, dtype=ts.dtype)
        curr_t, curr_y = ts[0], y0
        prev_t, prev_y = curr_t, curr_y
        ys = [y0]
        for out_t in ts[1:]:
            while curr_t < out_t:
                next_t = torch.minimum(curr_t + dt, ts[-1])
                prev_t, prev_y = curr_t, curr_y
                dW = bm(curr_t, next_t)
                curr_y = curr_y + sde.f(curr_t, curr_y) * (next_t - curr_t) + sde.g(curr_t, curr_y) * dW
                curr_t = next_t
            if curr_t == out_t or prev_t == curr_t:
                ys.append(curr_y)
            else:
                w = (out_t - prev_t) / (curr_t - prev_t)
                ys.append(prev_y + w * (curr_y - prev_y))
        return torch.stack(ys)

    mod("torchdyn")
    mod("torchdyn.core", NeuralODE=type("NeuralODE", (), {}))
    mod("torchdyn.numerics", Euler=type("Euler", (), {}), odeint=torchdyn_odeint)
    mod("torchdiffeq", odeint=torchdiffeq_odeint, odeint_adjoint=torchdiffeq_odeint)
    mod("torchsde", sdeint=torchsde_sdeint)


_install_standins()
sys.path.insert(0, "/root/reference")
