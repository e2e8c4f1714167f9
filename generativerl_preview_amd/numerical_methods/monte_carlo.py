"""Rejection sampler for t ~ pdf(t) on [x_min, x_max], used by the default ("maximum_likelihood")
score-matching weighting (grl/numerical_methods/monte_carlo.py:41-84).  Host-driven by design: its
trip count depends on the data; it is not part of the fused kernels."""
import math

import torch


class MonteCarloSampler:
    def __init__(self, unnormalized_pdf, x_min, x_max, device=torch.device("cpu")):
        self.unnormalized_pdf = unnormalized_pdf
        self.x_min, self.x_max, self.device = x_min, x_max, device
        self.uniform_dist = torch.distributions.uniform.Uniform(torch.tensor(x_min, device=device),
                                                                torch.tensor(x_max, device=device))

    def _sample(self, n):
        u = self.uniform_dist.sample((n,))
        pdf = self.unnormalized_pdf(u)
        keep = torch.rand(n, device=self.device) < pdf / torch.max(pdf)
        return u[keep]

    def sample(self, num_samples: int):
        got = torch.tensor([], device=self.device)
        ratio = 1.0
        while len(got) < num_samples:
            n = min(math.floor((num_samples - len(got)) * ratio), 2 ** 24)
            fresh = self._sample(n)
            ratio = n / fresh.shape[0]
            got = torch.cat([got, fresh])
        return got[:num_samples]
