from .q_network import DoubleQNetwork, QNetwork  # noqa: F401
from .value_network import VNetwork  # noqa: F401
