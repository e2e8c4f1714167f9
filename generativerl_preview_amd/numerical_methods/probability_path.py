"""Closed-form schedules of the Gaussian conditional probability paths
(grl/numerical_methods/probability_path.py:8-553), ``linear_vp_sde`` and ``gvp``.

Scalar functions of t only: they are evaluated on the host to build the per-step coefficient table
the fused integrator consumes (T-1 values per call) and, per row, to prepare the training batches.
The formulas are kept AS WRITTEN in the reference, including ``d_scale_dt`` of linear_vp_sde
omitting the beta_0 term (:342-343) and gvp's g^2 (:222-229)."""
import math

import torch

from ..config import as_config


class GaussianConditionalProbabilityPath:
    def __init__(self, config) -> None:
        self.config = as_config(config)
        self.type = self.config.type
        self.t_max = 1.0 if not hasattr(self.config, "t_max") else self.config.t_max
        assert self.type in ["diffusion", "vp_sde", "linear_vp_sde", "cosine_vp_sde", "general_ve_sde", "op_flow",
                             "linear", "gvp"], "Unknown type of Gaussian conditional probability path {}".format(self.type)
        if self.type == "linear_vp_sde":
            self.beta_0, self.beta_1 = float(self.config.beta_0), float(self.config.beta_1)

    def _unsupported(self, what):
        return NotImplementedError("{} for type {} is not implemented".format(what, self.type))

    def drift_coefficient(self, t):
        """f(t), probability_path.py:90-127"""
        if self.type == "linear_vp_sde":
            return -0.5 * (self.beta_0 + t * (self.beta_1 - self.beta_0))
        if self.type == "gvp":
            return -0.5 * math.pi * torch.tan(math.pi * t / 2.0)
        raise self._unsupported("Drift coefficient term")

    def drift(self, t, x=None):
        """f(t) x, probability_path.py:129-159"""
        return torch.einsum("i...,i->i...", x, self.drift_coefficient(t))

    def diffusion_squared(self, t):
        """g^2(t), probability_path.py:199-233"""
        if self.type == "linear_vp_sde":
            return self.beta_0 + t * (self.beta_1 - self.beta_0)
        if self.type == "gvp":
            s = torch.sin(math.pi * t * 0.5)
            return math.pi * s + s * s * torch.tan(math.pi * t * 0.5)
        raise self._unsupported("Diffusion term")

    def diffusion(self, t):
        """g(t), probability_path.py:161-197"""
        return torch.sqrt(self.diffusion_squared(t))

    def scale(self, t):
        """s(t), probability_path.py:235-263"""
        if self.type == "linear_vp_sde":
            return torch.exp(-0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0)
        if self.type == "gvp":
            return torch.cos(0.5 * math.pi * t)
        raise self._unsupported("Scale factor")

    def log_scale(self, t):
        """probability_path.py:265-293"""
        if self.type == "linear_vp_sde":
            return -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        if self.type == "gvp":
            return torch.log(torch.cos(0.5 * math.pi * t))
        raise self._unsupported("Log scale factor")

    def d_log_scale_dt(self, t):
        """probability_path.py:295-323"""
        if self.type == "linear_vp_sde":
            return -0.5 * t * (self.beta_1 - self.beta_0)
        if self.type == "gvp":
            return -0.5 * math.pi * torch.tan(0.5 * math.pi * t)
        raise self._unsupported("Time derivative of the log scale factor")

    def d_scale_dt(self, t):
        """s'(t) as written (no beta_0 term for linear_vp_sde), probability_path.py:325-353"""
        if self.type == "linear_vp_sde":
            return -0.5 * t * (self.beta_1 - self.beta_0) * self.scale(t)
        if self.type == "gvp":
            return -0.5 * math.pi * torch.sin(0.5 * math.pi * t)
        raise self._unsupported("Time derivative of the scale factor")

    def std(self, t):
        """sigma(t), probability_path.py:355-388"""
        if self.type == "linear_vp_sde":
            return torch.sqrt(1.0 - self.scale(t) ** 2)
        if self.type == "gvp":
            return torch.sin(0.5 * math.pi * t)
        raise self._unsupported("Standard deviation")

    def d_std_dt(self, t):
        """probability_path.py:390-413"""
        if self.type == "linear_vp_sde":
            return -self.d_scale_dt(t) * self.scale(t) / self.std(t)
        if self.type == "gvp":
            return 0.5 * math.pi * torch.cos(0.5 * math.pi * t)
        raise self._unsupported("Time derivative of standard deviation")

    def covariance(self, t):
        """Sigma(t), probability_path.py:415-448"""
        if self.type == "linear_vp_sde":
            return 1.0 - self.scale(t) ** 2
        if self.type == "gvp":
            s = torch.sin(0.5 * math.pi * t)
            return s * s
        raise self._unsupported("Covariance")

    def d_covariance_dt(self, t):
        """probability_path.py:450-477"""
        if self.type == "linear_vp_sde":
            return -2.0 * self.scale(t) * self.d_scale_dt(t)
        if self.type == "gvp":
            return math.pi * torch.sin(math.pi * t * 0.5) * torch.cos(math.pi * t * 0.5)
        raise self._unsupported("Time derivative of covariance")

    def HalfLogSNR(self, t):
        """probability_path.py:479-498"""
        if self.type == "linear_vp_sde":
            return self.log_scale(t) - 0.5 * torch.log(1.0 - self.scale(t) ** 2)
        raise self._unsupported("Half-logSNR")

    def InverseHalfLogSNR(self, HalfLogSNR):
        """probability_path.py:500-538"""
        if self.type == "linear_vp_sde":
            num = 2.0 * torch.logaddexp(-2.0 * HalfLogSNR, torch.zeros((1,)).to(HalfLogSNR))
            den = torch.sqrt(self.beta_0 ** 2 + (self.beta_1 - self.beta_0) * num) + self.beta_0
            return num / den
        raise self._unsupported("Inverse function of half-logSNR")


class ConditionalProbabilityPath:
    """Constant-sigma path of the conditional flow models, probability_path.py:541-553."""

    def __init__(self, config) -> None:
        self.config = as_config(config)

    def std(self, t):
        # (a fill, not a host-to-device copy: the training step stays capturable into a HIP graph)
        return torch.full((), float(self.config.sigma), device=t.device)
