from .losses import fused_loss, prepare_flow_matching, prepare_icfm, prepare_score_matching  # noqa: F401
