"""grl/algorithms: only the loss-bearing modules of the hot path's callers; the training
loops, datasets and logging around them stay out of scope."""
from .gmpo import GMPOCritic, GMPOPolicy  # noqa: F401
from .qgpo import QGPOCritic  # noqa: F401
