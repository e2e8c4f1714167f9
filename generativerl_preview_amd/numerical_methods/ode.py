"""grl/numerical_methods/ode.py:6-25 -- container for an ODE dx = f(t, x) dt."""


class ODE:
    def __init__(self, drift=None, ode_func=None, t_max: float = 1.0):
        self.drift = drift
        self.ode_func = ode_func
        self.t_max = t_max
