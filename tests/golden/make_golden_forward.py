"""Golden vectors for the encoding direction (data -> latent), produced by the REFERENCE's own
``forward_sample_process`` (grl/generative_models/diffusion_model/diffusion_model.py:890-983 with
DiffusionProcess.forward_ode / forward_sde, diffusion_process.py:559-740; and
conditional_flow_model/independent_conditional_flow_model.py:772-846).  Container-only script:
    python tests/golden/make_golden_forward.py        ->  tests/golden/forward.npz
Weights are the same seeded ones as make_golden.py (golden_lib); the integrators are the stand-ins of ref_shims.py
(stepping parity unpinned, as for every sampling golden).  The grid starts at t = 0.05: at t = 0 the noise-head
closure divides by std(0) = 0 in the reference too."""
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

torch.set_num_threads(8)


def build(name):
    kind, cfg = gl.model_zoo()[name]
    cfg = EasyDict(cfg)
    m = DiffusionModel(cfg) if kind == "diffusion" else IndependentConditionalFlowModel(cfg)
    key_shapes = [(k, list(v.shape)) for k, v in m.state_dict().items()]
    w = gl.load_weights(name, key_shapes)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    return kind, cfg, m


out = {}
RAISES = []
N_ROWS = 5
for name in ("small_vp_noise", "small_vp_score", "small_vp_velocity", "small_vp_data", "small_gvp_velocity",
             "small_uncond_noise", "small_icfm", "swiss_noise", "icfm_cond"):
    kind, cfg, m = build(name)
    D = cfg.x_size
    S = cfg.model.args.backbone.args.get("condition_dim", None)
    gen = torch.Generator().manual_seed(gl._seed_for(name, 55))
    x = torch.randn(N_ROWS, D, generator=gen) * 0.7
    cond = torch.randn(N_ROWS, S, generator=gen) if S else None
    out[name + "/x"] = x.numpy()
    if S:
        out[name + "/cond"] = cond.numpy()
    t_span = torch.linspace(0.05, 1.0, 8)
    try:
        traj = m.forward_sample_process(x=x, t_span=t_span, condition=cond)
        out[name + "/ode"] = traj.detach().numpy()
        final = m.forward_sample(x=x, t_span=t_span, condition=cond)
        assert torch.equal(final, traj[-1])
        print(name, "ode", tuple(traj.shape), float(traj[-1].abs().max()))
    except Exception as e:  # noqa: BLE001 -- record reference failures instead of hiding them
        print("reference raises for", name, "ode", type(e).__name__, str(e)[:100])
        RAISES.append("%s/ode: %s" % (name, type(e).__name__))
    if kind != "diffusion":
        continue
    dt = 0.25
    t_sde = torch.tensor([0.05, 0.3, 0.55, 0.8, 1.0])
    rs = np.random.RandomState(23)
    dW = (rs.randn(4, N_ROWS, D) * np.sqrt(dt)).astype(np.float32)

    class BM:
        def __init__(self):
            self.k = 0

        def __call__(self, ta, tb):
            w = torch.from_numpy(dW[self.k])
            self.k += 1
            return w

    try:
        traj = m.forward_sample_process(x=x, t_span=t_sde, condition=cond, solver_config=EasyDict(
            type="SDESolver", args=dict(library="torchsde", dt=dt, bm=BM())))
        out[name + "/sde"], out[name + "/sde_dW"] = traj.detach().numpy(), dW
        print(name, "sde", tuple(traj.shape), float(traj[-1].abs().max()))
    except Exception as e:  # noqa: BLE001
        print("reference raises for", name, "sde", type(e).__name__, str(e)[:100])
        RAISES.append("%s/sde: %s" % (name, type(e).__name__))
out["raises"] = np.array(sorted(set(RAISES)))
np.savez_compressed(os.path.join(HERE, "forward.npz"), **out)
print("wrote forward.npz", sum(a.nbytes for a in out.values()) // 1024, "KiB")
