from .diffusion_model import DiffusionModel  # noqa: F401
from .energy_conditional_diffusion_model import EnergyConditionalDiffusionModel, EnergyGuidance  # noqa: F401
