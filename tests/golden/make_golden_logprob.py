"""Golden vectors for the likelihood entry points, produced by the REFERENCE: ``compute_likelihood``
(grl/generative_models/metric.py:14-139) behind ``DiffusionModel.log_prob`` (diffusion_model.py:1125-1159), and
``IndependentConditionalFlowModel.log_prob`` (independent_conditional_flow_model.py:515-631).  Container-only script:
    python tests/golden/make_golden_logprob.py        ->  tests/golden/logprob.npz
The integrator is the Euler stand-in of ref_shims.py (torchdiffeq is absent; stepping parity unpinned), the composite
drift, both trace computations and the prior term are the reference's code.  The Hutchinson probes
(``torch.randn_like`` inside the drift, one per step) are captured by replaying the generator."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import golden_lib as gl  # noqa: E402
from ref_shims import EasyDict  # noqa: E402
from grl.generative_models.diffusion_model.diffusion_model import DiffusionModel  # noqa: E402
from grl.generative_models.conditional_flow_model.independent_conditional_flow_model import (  # noqa: E402
    IndependentConditionalFlowModel,
)
from grl.generative_models.metric import compute_likelihood  # noqa: E402

torch.set_num_threads(8)
out = {}
N_ROWS = 6
for name in ("small_vp_noise", "small_vp_score", "small_gvp_velocity", "small_uncond_noise", "small_icfm"):
    kind, cfg = gl.model_zoo()[name]
    cfg = EasyDict(cfg)
    m = DiffusionModel(cfg) if kind == "diffusion" else IndependentConditionalFlowModel(cfg)
    key_shapes = [(k, list(v.shape)) for k, v in m.state_dict().items()]
    m.load_state_dict({k: torch.from_numpy(v) for k, v in gl.load_weights(name, key_shapes).items()})
    D = cfg.x_size
    S = cfg.model.args.backbone.args.get("condition_dim", None)
    gen = torch.Generator().manual_seed(gl._seed_for(name, 88))
    x = torch.randn(N_ROWS, D, generator=gen) * 0.8
    cond = torch.randn(N_ROWS, S, generator=gen) if S else None
    pre = name + "/"
    out[pre + "x"] = x.numpy()
    if S:
        out[pre + "cond"] = cond.numpy()
    grid = torch.linspace(1e-3, 1.0, 41) if kind == "diffusion" else torch.linspace(0.0, 1.0, 41)
    out[pre + "grid"] = grid.numpy()
    seed = gl._seed_for(name, 89)

    def call(hutch, t):
        if kind == "diffusion":
            return compute_likelihood(model=m, x=x, t=t, condition=cond, using_Hutchinson_trace_estimator=hutch)
        return m.log_prob(x_1=x, condition=cond, t=t, using_Hutchinson_trace_estimator=hutch)

    out[pre + "exact"] = call(False, grid).detach().numpy()
    torch.manual_seed(seed)
    out[pre + "probes"] = torch.stack([torch.randn(N_ROWS, D) for _ in range(len(grid) - 1)]).numpy()
    torch.manual_seed(seed)
    out[pre + "hutchinson"] = call(True, grid).detach().numpy()
    print(name, out[pre + "exact"][:3], out[pre + "hutchinson"][:3])
    if name == "small_uncond_noise":   # the default grid of DiffusionModel.log_prob: linspace(1e-3, 1, 1000)
        torch.manual_seed(seed)
        out[pre + "probes_default"] = torch.stack([torch.randn(N_ROWS, D) for _ in range(999)]).numpy()
        torch.manual_seed(seed)
        out[pre + "log_prob_default"] = m.log_prob(x=x).detach().numpy()
        print(name, "default grid", out[pre + "log_prob_default"][:3])
np.savez_compressed(os.path.join(HERE, "logprob.npz"), **out)
print("wrote logprob.npz", sum(a.nbytes for a in out.values()) // 1024, "KiB")
