"""Golden vectors for sampling with pinned components, produced by the REFERENCE's own
``sample_forward_process_with_fixed_x`` (grl/generative_models/diffusion_model/diffusion_model.py:521-859 and
energy_conditional_diffusion_model.py:669-1036).  Container-only script:
    python tests/golden/make_golden_fixed_x.py        ->  tests/golden/fixed_x.npz
Weights are the same seeded ones as make_golden.py (golden_lib).

The reference draws fresh Gaussian noise inside the drift (``direct_sample`` -> ``torch.randn_like``): one draw per
drift evaluation (ODE), a drift draw then a diffusion draw per step (SDE; the diffusion coefficient does not depend on x,
so its draw only advances the generator).  The draws are captured by replaying the generator from the same seed in the
same order (x_T first when x_0 is None), and stored as ``noise`` so that the oracle and the HIP path can consume them.
The integrators are the stand-ins of ref_shims.py (stepping parity unpinned, as for every sampling golden)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import golden_lib as gl  # noqa: E402
from ref_shims import EasyDict  # noqa: E402
from grl.generative_models.diffusion_model.diffusion_model import DiffusionModel  # noqa: E402
from grl.generative_models.diffusion_model.energy_conditional_diffusion_model import (  # noqa: E402
    EnergyConditionalDiffusionModel,
)

torch.set_num_threads(8)


def build(name):
    kind, cfg = gl.model_zoo()[name]
    cfg = EasyDict(cfg)
    m = DiffusionModel(cfg) if kind == "diffusion" else EnergyConditionalDiffusionModel(cfg, energy_model=None)
    key_shapes = [(k, list(v.shape)) for k, v in m.state_dict().items()]
    w = gl.load_weights(name, key_shapes)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    return kind, cfg, m


out = {}
RAISES = []
N_ROWS = 4
for name in ("small_vp_noise", "small_vp_score", "small_gvp_velocity", "small_uncond_noise", "small_energy", "qgpo_noise"):
    kind, cfg, m = build(name)
    D = cfg.x_size
    S = cfg.model.args.backbone.args.get("condition_dim", None)
    gen = torch.Generator().manual_seed(gl._seed_for(name, 91))
    x0 = torch.randn(N_ROWS, D, generator=gen)
    cond = torch.randn(N_ROWS, S, generator=gen) if S else None
    fixed_x = torch.randn(N_ROWS, D, generator=gen) * 0.5
    mask = (torch.rand(N_ROWS, D, generator=gen) < 0.5).float()
    mask[0, :] = 1.0          # a row with nothing pinned
    mask[1, :] = 0.0          # a row with everything pinned
    if D > 1:
        mask[2, 0] = 0.35     # a fractional mask: the reference's arithmetic blends
    out[name + "/x0"], out[name + "/fixed_x"], out[name + "/mask"] = x0.numpy(), fixed_x.numpy(), mask.numpy()
    if S:
        out[name + "/cond"] = cond.numpy()
    t_span = torch.linspace(0.0, 1.0, 7)
    n_steps = len(t_span) - 1
    scales = (0.0, 1.0) if kind == "energy" else (None,)
    # (tag, batch_size, use_x0)
    for tag, B, use_x0 in (("x0", None, True), ("B2_x0", 2, True), ("B3_draw", 3, False)):
        rep = 1 if B is None else B
        for gs in scales:
            seed = gl._seed_for(name, 300 + rep)
            torch.manual_seed(seed)
            xT = None if use_x0 else torch.randn(rep * N_ROWS, D)
            noise = torch.stack([torch.randn(rep * N_ROWS, D) for _ in range(n_steps)])
            kw = dict(fixed_x=fixed_x, fixed_mask=mask, t_span=t_span, batch_size=B, condition=cond,
                      x_0=x0 if use_x0 else None)
            if gs is not None:
                kw["guidance_scale"] = gs
            torch.manual_seed(seed)
            traj = m.sample_forward_process_with_fixed_x(**kw)
            key = "%s/ode_%s%s" % (name, tag, "" if gs is None else "_g%g" % gs)
            out[key] = traj.detach().numpy()
            out["%s/ode_%s_noise" % (name, tag)] = noise.numpy()
            if xT is not None:
                out["%s/ode_%s_xT" % (name, tag)] = xT.numpy()
            print(key, tuple(traj.shape), float(traj[-1].abs().max()))
    # Euler-Maruyama (the reference marks this branch "TODO: validate"; recorded where it runs)
    dt = 0.25
    t_sde = torch.linspace(0.0, 1.0, 5)
    rs = np.random.RandomState(11)
    dW = (rs.randn(4, N_ROWS, D) * np.sqrt(dt)).astype(np.float32)

    class BM:
        def __init__(self):
            self.k = 0

        def __call__(self, ta, tb):
            w = torch.from_numpy(dW[self.k])
            self.k += 1
            return w

    for gs in scales:
        seed = gl._seed_for(name, 411)
        torch.manual_seed(seed)
        draws = [torch.randn(N_ROWS, D) for _ in range(2 * 4)]     # drift draw, diffusion draw, per step
        noise = torch.stack(draws[0::2])
        kw = dict(fixed_x=fixed_x, fixed_mask=mask, t_span=t_sde, x_0=x0, condition=cond,
                  solver_config=EasyDict(type="SDESolver", args=dict(library="torchsde", dt=dt, bm=BM())))
        if gs is not None:
            kw["guidance_scale"] = gs
        torch.manual_seed(seed)
        try:
            traj = m.sample_forward_process_with_fixed_x(**kw)
        except Exception as e:  # noqa: BLE001 -- record reference failures instead of hiding them
            print("reference raises for", name, "sde", gs, type(e).__name__, str(e)[:100])
            RAISES.append("%s/sde%s: %s" % (name, "" if gs is None else "_g%g" % gs, type(e).__name__))
            continue
        key = "%s/sde%s" % (name, "" if gs is None else "_g%g" % gs)
        out[key] = traj.detach().numpy()
        out[name + "/sde_noise"], out[name + "/sde_dW"] = noise.numpy(), dW
        print(key, tuple(traj.shape), float(traj[-1].abs().max()))
out["raises"] = np.array(sorted(set(RAISES)))
np.savez_compressed(os.path.join(HERE, "fixed_x.npz"), **out)
print("wrote fixed_x.npz", sum(a.nbytes for a in out.values()) // 1024, "KiB")
