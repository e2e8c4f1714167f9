"""StochasticProcess of the conditional flow models (grl/generative_models/stochastic_process.py:14-129):
x_t = (1 - t) x0 + t x1 + sigma * eps, velocity x1 - x0."""
import torch


class StochasticProcess:
    def __init__(self, path, t_max: float = 1.0) -> None:
        self.path = path
        self.t_max = t_max

    def mean(self, t, x0, x1, condition=None):
        if x0.dim() > t.dim():
            t = t[(...,) + (None,) * (x0.dim() - t.dim())].expand(x0.shape)
        return x0 * (1 - t) + x1 * t

    def std(self, t, x0, x1, condition=None):
        s = self.path.std(t)
        if x0.dim() > s.dim():
            s = s[(...,) + (None,) * (x0.dim() - s.dim())].expand(x0.shape)
        return s

    def velocity(self, t, x0, x1, condition=None):
        return x1 - x0

    def direct_sample(self, t, x0, x1, condition=None):
        return self.mean(t, x0, x1, condition) + self.std(t, x0, x1, condition) * torch.randn_like(x0)
