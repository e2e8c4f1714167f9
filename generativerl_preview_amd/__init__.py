"""MI355X-native implementation of GenerativeRL's generative-policy hot path (batched action
sampling + score/flow-matching training step) behind the reference's own model/solver API.
Host code is a thin mirror of that API; the arithmetic runs in libgrlhip.so (include/grlhip.h)."""
from . import _lib  # noqa: F401
from .config import AttrDict, as_config  # noqa: F401
from .generative_models import (DiffusionModel, EnergyConditionalDiffusionModel,  # noqa: F401
                                IndependentConditionalFlowModel, get_generative_model)
from .numerical_methods.numerical_solvers import ODESolver, SDESolver, get_solver  # noqa: F401

__all__ = ["DiffusionModel", "EnergyConditionalDiffusionModel", "IndependentConditionalFlowModel",
           "get_generative_model", "ODESolver", "SDESolver", "get_solver", "AttrDict", "as_config"]
