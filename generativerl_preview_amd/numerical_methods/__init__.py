from .ode import ODE  # noqa: F401
from .sde import SDE  # noqa: F401
