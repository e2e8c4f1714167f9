"""grl/numerical_methods/sde.py:6-27 -- container for an SDE dx = f(t, x) dt + g(t, x) dw."""


class SDE:
    def __init__(self, drift=None, diffusion=None, t_max: float = 1.0):
        self.drift = drift
        self.diffusion = diffusion
        self.t_max = t_max
