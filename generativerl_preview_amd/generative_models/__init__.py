"""Generative-model registry (grl/generative_models/__init__.py:1-20), hot-path subset."""
from .conditional_flow_model import IndependentConditionalFlowModel
from .diffusion_model import DiffusionModel, EnergyConditionalDiffusionModel


def get_generative_model(name: str):
    if name.lower() not in GENERATIVE_MODELS:
        raise ValueError("Unknown activation function {}".format(name))
    return GENERATIVE_MODELS[name.lower()]


GENERATIVE_MODELS = {
    "DiffusionModel".lower(): DiffusionModel,
    "EnergyConditionalDiffusionModel".lower(): EnergyConditionalDiffusionModel,
    "IndependentConditionalFlowModel".lower(): IndependentConditionalFlowModel,
}
