from .independent_conditional_flow_model import IndependentConditionalFlowModel  # noqa: F401
