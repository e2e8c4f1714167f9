"""DiffusionProcess (grl/generative_models/diffusion_process.py:13-978), tensor states.

Schedule accessors broadcast the scalar schedule to ``x.shape`` exactly like the reference
(``[..., None].expand``); ``reverse_ode`` / ``reverse_sde`` return ODE/SDE containers whose drift is a
``FusedDrift`` (generative_models/fused.py) when ``function`` is an IntrinsicModel, so the solvers can
run the whole integration in one HIP launch."""
import torch

from ..numerical_methods.ode import ODE
from ..numerical_methods.sde import SDE
from .fused import FusedDrift
from .intrinsic_model import IntrinsicModel


def _bc(v, x):
    if x is not None and x.dim() > v.dim():
        return v[(...,) + (None,) * (x.dim() - v.dim())].expand(x.shape)
    return v


class DiffusionProcess:
    def __init__(self, path, t_max: float = 1.0):
        self.path = path
        self.t_max = t_max

    def drift(self, t, x, condition=None):            # :31-73
        return _bc(self.path.drift_coefficient(t), x) * x

    def drift_coefficient(self, t, x=None, condition=None):   # :75-117
        return _bc(self.path.drift_coefficient(t), x)

    def diffusion(self, t, x=None, condition=None):   # :119-161
        return _bc(self.path.diffusion(t), x)

    def diffusion_squared(self, t, x=None, condition=None):   # :163-205
        return _bc(self.path.diffusion_squared(t), x)

    def scale(self, t, x=None, condition=None):       # :207-249
        return _bc(self.path.scale(t), x)

    def log_scale(self, t, x=None, condition=None):   # :251-293
        return _bc(self.path.log_scale(t), x)

    def std(self, t, x=None, condition=None):         # :295-337
        return _bc(self.path.std(t), x)

    def covariance(self, t, x=None, condition=None):  # :339-381
        return _bc(self.path.covariance(t), x)

    def velocity(self, t, x, condition=None, noise=None):     # :383-459
        if noise is None:
            noise = torch.randn_like(x)
        return _bc(self.path.d_scale_dt(t), x) * x + _bc(self.path.d_std_dt(t), x) * noise

    def HalfLogSNR(self, t, x=None, condition=None):
        return _bc(self.path.HalfLogSNR(t), x)

    def InverseHalfLogSNR(self, HalfLogSNR, x=None, condition=None):
        return self.path.InverseHalfLogSNR(HalfLogSNR)

    def direct_sample(self, t, x, condition=None):    # :939-956
        return self.scale(t, x) * x + self.std(t, x) * torch.randn_like(x)

    def direct_sample_and_return_noise(self, t, x, condition=None):
        noise = torch.randn_like(x)
        return self.scale(t, x) * x + self.std(t, x) * noise, noise

    def _fused(self, kind, function, function_type, condition, reverse_path=None, T=1.0):
        guidance, scale = None, 0.0
        if isinstance(function, GuidedScore):
            guidance, scale, function_type, function = function.guidance, function.scale, function.model_type, function.model
        if not isinstance(function, IntrinsicModel):
            raise NotImplementedError("reverse_ode/reverse_sde need the model's IntrinsicModel (or a GuidedScore) as "
                                      "`function`; arbitrary Python callables cannot enter the fused integrator")
        return FusedDrift(kind, function, function_type, self.path, reverse_path, condition, guidance, scale,
                          float(T))

    def reverse_ode(self, function, function_type, condition=None, T=torch.tensor(1.0)):   # :853-919
        return ODE(drift=self._fused("reverse_ode", function, function_type, condition, None, T))

    def reverse_sde(self, function, function_type, reverse_diffusion_function=None,
                    reverse_diffusion_squared_function=None, condition=None, T=torch.tensor(1.0)):  # :742-851
        rproc = getattr(reverse_diffusion_function, "__self__", None)
        if not isinstance(rproc, DiffusionProcess):
            raise NotImplementedError("reverse_diffusion_function must be a DiffusionProcess.diffusion bound method")
        d = self._fused("reverse_sde", function, function_type, condition, rproc.path, T)
        return SDE(drift=d, diffusion=d)


    def forward_ode(self, function, function_type, condition=None, T=torch.tensor(1.0)):   # :672-740
        return ODE(drift=self._fused("forward_ode", function, function_type, condition, None, T))

    def forward_sde(self, function, function_type, forward_diffusion_function=None,
                    forward_diffusion_squared_function=None, condition=None, T=torch.tensor(1.0)):  # :559-670
        fproc = getattr(forward_diffusion_function, "__self__", None)
        if not isinstance(fproc, DiffusionProcess):
            raise NotImplementedError("forward_diffusion_function must be a DiffusionProcess.diffusion bound method")
        d = self._fused("forward_sde", function, function_type, condition, fproc.path, T)
        return SDE(drift=d, diffusion=d)


class GuidedScore:
    """score(t,x,c) + guidance_scale * dE/dx (energy_conditional_diffusion_model.py:1109-1133) as a
    descriptor: the model head, the energy net and the scale."""

    def __init__(self, model, model_type, guidance, scale):
        self.model, self.model_type, self.guidance, self.scale = model, model_type, guidance, float(scale)
