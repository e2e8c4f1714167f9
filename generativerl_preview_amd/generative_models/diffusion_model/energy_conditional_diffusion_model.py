"""EnergyConditionalDiffusionModel (QGPO), grl/generative_models/diffusion_model/
energy_conditional_diffusion_model.py:29-1394: a diffusion model plus an energy-guidance net; guided
sampling integrates score + guidance_scale * dE/dx, with the Q-gradient evaluated inside the same
fused step (csrc/engine_bf16.hip.inc ``guide_grad``; fp32: the batch engine's gradient program)."""
import torch
import torch.nn as nn

from ...config import as_config
from ..diffusion_process import GuidedScore
from ..intrinsic_model import IntrinsicModel
from .diffusion_model import DiffusionModel


class EnergyGuidance(nn.Module):
    """energy_conditional_diffusion_model.py:29-92"""

    def __init__(self, config):
        super().__init__()
        self.config = as_config(config)
        self.model = IntrinsicModel(self.config)

    def forward(self, t, x, condition=None):
        return self.model(t, x, condition)

    def calculate_energy_guidance(self, t, x, condition=None, guidance_scale: float = 1.0):
        """guidance_scale * d/dx sum E(t, x, c) (:67-92), one fused forward+input-gradient launch."""
        m = self.model
        if not hasattr(m, "_io_dims"):
            m.set_io_dims(x.shape[-1], 0 if condition is None else condition.shape[-1])
        t = t.expand(x.shape[0]) if t.dim() == 0 else t
        return m.plan.energy_grad(m.packed(), t, x, condition, guidance_scale)


class EnergyConditionalDiffusionModel(DiffusionModel):
    def __init__(self, config, energy_model=None) -> None:
        config = as_config(config)
        super().__init__(config)
        self.energy_model = energy_model
        self.energy_guidance = EnergyGuidance(config.energy_guidance)
        x_dim = config.x_size if isinstance(config.x_size, int) else int(torch.Size(config.x_size).numel())
        bb = config.model.args.backbone.args
        self.energy_guidance.model.set_io_dims(x_dim, bb.condition_dim if hasattr(bb, "condition_dim") else 0)

    def get_type(self):
        return "EnergyConditionalDiffusionModel"

    def _function(self, guidance_scale):
        # the reference always integrates the guided score, also for guidance_scale == 0
        # (energy_conditional_diffusion_model.py:406-435; qgpo.py:479)
        gs = 1.0 if guidance_scale is None else guidance_scale
        return GuidedScore(self.model, self.model_type, self.energy_guidance.model, gs), "score_function"

    def _dpm_functions(self, guidance_scale):
        from ..fused import AffineModelFn
        gs = 1.0 if guidance_scale is None else guidance_scale
        kw = dict(guidance=self.energy_guidance.model, guidance_scale=gs)
        return (AffineModelFn("noise", self.model, self.model_type, self.path,
                              call=lambda t, x, c: self.noise_function_with_energy_guidance(t, x, c, gs), **kw),
                AffineModelFn("data", self.model, self.model_type, self.path,
                              call=lambda t, x, c: self.data_prediction_function_with_energy_guidance(t, x, c, gs), **kw))

    def noise_function_with_energy_guidance(self, t, x, condition=None, guidance_scale: float = 1.0):
        """:1235-1260"""
        return -self.score_function_with_energy_guidance(t, x, condition, guidance_scale) * self.diffusion_process.std(t, x)

    def data_prediction_function_with_energy_guidance(self, t, x, condition=None, guidance_scale: float = 1.0):
        """:1281-1306 (as written: -std * x)"""
        dp = self.diffusion_process
        return (-dp.std(t, x) * x + dp.covariance(t, x)
                * self.score_function_with_energy_guidance(t, x, condition, guidance_scale)) / dp.scale(t, x)

    def sample(self, t_span=None, batch_size=None, x_0=None, condition=None, guidance_scale: float = 1.0,
               with_grad: bool = False, solver_config=None, **kw):
        """:168-213"""
        return super().sample_forward_process(t_span=t_span, batch_size=batch_size, x_0=x_0, condition=condition,
                                              with_grad=with_grad, solver_config=solver_config, _final_only=True,
                                              _guidance_scale=guidance_scale, **kw)

    def sample_forward_process(self, t_span=None, batch_size=None, x_0=None, condition=None,
                               guidance_scale: float = 1.0, with_grad: bool = False, solver_config=None, **kw):
        """:257-615"""
        return super().sample_forward_process(t_span=t_span, batch_size=batch_size, x_0=x_0, condition=condition,
                                              with_grad=with_grad, solver_config=solver_config,
                                              _guidance_scale=guidance_scale, **kw)

    def sample_with_fixed_x(self, fixed_x, fixed_mask, t_span=None, batch_size=None, x_0=None, condition=None,
                            guidance_scale: float = 1.0, with_grad: bool = False, solver_config=None, **kw):
        """:617-667"""
        return super().sample_forward_process_with_fixed_x(
            fixed_x=fixed_x, fixed_mask=fixed_mask, t_span=t_span, batch_size=batch_size, x_0=x_0, condition=condition,
            with_grad=with_grad, solver_config=solver_config, _final_only=True, _guidance_scale=guidance_scale, **kw)

    def sample_forward_process_with_fixed_x(self, fixed_x, fixed_mask, t_span=None, batch_size=None, x_0=None,
                                            condition=None, guidance_scale: float = 1.0, with_grad: bool = False,
                                            solver_config=None, **kw):
        """:669-1036 -- the guided score integrated with pinned components"""
        return super().sample_forward_process_with_fixed_x(
            fixed_x=fixed_x, fixed_mask=fixed_mask, t_span=t_span, batch_size=batch_size, x_0=x_0, condition=condition,
            with_grad=with_grad, solver_config=solver_config, _guidance_scale=guidance_scale, **kw)

    def sample_with_fixed_x_without_energy_guidance(self, fixed_x, fixed_mask, t_span=None, batch_size=None, x_0=None,
                                                    condition=None, with_grad: bool = False, solver_config=None):
        """:1038-1083"""
        return self.sample_with_fixed_x(fixed_x=fixed_x, fixed_mask=fixed_mask, t_span=t_span, batch_size=batch_size,
                                        x_0=x_0, condition=condition, guidance_scale=0.0, with_grad=with_grad,
                                        solver_config=solver_config)

    def sample_without_energy_guidance(self, t_span=None, batch_size=None, x_0=None, condition=None,
                                       with_grad: bool = False, solver_config=None):
        """:215-255"""
        return self.sample(t_span=t_span, batch_size=batch_size, x_0=x_0, condition=condition, guidance_scale=0.0,
                           with_grad=with_grad, solver_config=solver_config)

    def score_function_with_energy_guidance(self, t, x, condition=None, guidance_scale: float = 1.0):
        """:1109-1133"""
        from ..fused import _head_to_score
        net = self.model(t, x, condition)
        p, q = _head_to_score(self.model_type, self.path, t)
        return p[:, None] * net + q[:, None] * x + self.energy_guidance.calculate_energy_guidance(
            t, x, condition, guidance_scale)

    def energy_guidance_loss(self, x, condition=None, t=None, noise=None, energy=None):
        """:1308-1394, contrastive energy prediction.  x: (B, K, D) support actions per state, condition: (B, C).
        The energy net's forward, the group softmax cross-entropy and the backward run in one HIP library call;
        ``self.energy_model`` (any module, evaluated without grad as in the reference) supplies the energies unless
        ``energy`` (B, K) is given.  Supersets: ``t`` (B,) and ``noise`` (B, K, D) pin the random draws, made in
        the reference's order otherwise (t first, then the noise of direct_sample)."""
        from ..model_functions.losses import fused_cep_loss
        if not torch.is_tensor(x) or x.dim() != 3:
            raise NotImplementedError("energy_guidance_loss: x must be a (B, K, D) tensor (dict-typed states are out of scope)")
        B, K, D = x.shape
        if K == 1:  # the reference's squeeze(dim=-1) drops the group axis and LogSoftmax(dim=1) raises (:1380, :1389)
            raise IndexError("Dimension out of range (expected to be in range of [-1, 0], but got 1)")
        eps = 1e-3
        if t is None:
            t = torch.rand((B,), device=x.device) * (1.0 - eps) + eps
        x_flat = x.reshape(B * K, D)
        cond_rep = None if condition is None else condition.unsqueeze(1).expand(B, K, *condition.shape[1:]).reshape(
            B * K, *condition.shape[1:])
        if energy is None:
            with torch.no_grad():
                energy = self.energy_model(x_flat, cond_rep) if cond_rep is not None else self.energy_model(x_flat)
        energy = energy.detach().reshape(B, K)
        if noise is None:
            noise = torch.randn_like(x)
        t_rep = t.unsqueeze(1).expand(B, K)
        pth = self.path
        x_t = pth.scale(t_rep)[..., None] * x + pth.std(t_rep)[..., None] * noise
        return fused_cep_loss(self.energy_guidance.model, t_rep.reshape(B * K), x_t.reshape(B * K, D), cond_rep,
                              (energy * self.alpha).reshape(B * K), K)
