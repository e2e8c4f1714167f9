from .q_network import DoubleQNetwork, QNetwork  # noqa: F401
