"""Golden vectors for the DPM-Solver path, produced by the REFERENCE's own
DPMSolver (grl/numerical_methods/numerical_solvers/dpm_solver.py -- reference code, not a third-party
package) driven through the reference model classes' sample_forward_process.  Container-only script:
    python tests/golden/make_golden_dpm.py        ->  tests/golden/dpm.npz
Weights and inputs are the same seeded ones as make_golden.py (golden_lib)."""
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

# (order, steps, type, method, solver_type, skip_type, denoise)
CASES = [
    (2, 17, "dpm_solver", "singlestep", "dpm_solver", "time_uniform", False),   # the QGPO benchmark configuration
    (1, 10, "dpm_solver", "singlestep", "dpm_solver", "time_uniform", False),
    (3, 10, "dpm_solver", "singlestep", "dpm_solver", "time_uniform", False),
    (2, 10, "dpm_solver++", "singlestep", "dpm_solver", "time_uniform", False),
    (3, 9, "dpm_solver++", "singlestep", "taylor", "time_quadratic", False),
    (2, 8, "dpm_solver", "singlestep", "taylor", "time_quadratic", True),
    (3, 11, "dpm_solver", "singlestep", "taylor", "time_uniform", False),
    # the reference itself raises on these two (skip_type "logSNR": exp() of a float; method "singlestep_fixed":
    # unbound `order`); kept so that the manifest of raising cases is reproducible
    (2, 10, "dpm_solver", "singlestep", "dpm_solver", "logSNR", False),
    (2, 10, "dpm_solver", "singlestep_fixed", "dpm_solver", "time_uniform", False),
]
RAISES = []


def case_key(c):
    return "o%d_n%d_%s_%s_%s_%s_d%d" % (c[0], c[1], c[2].replace("+", "p"), c[3], c[4], c[5], int(c[6]))


def build(name):
    kind, cfg = gl.model_zoo()[name]
    cfg = EasyDict(cfg)
    m = DiffusionModel(cfg) if kind == "diffusion" else EnergyConditionalDiffusionModel(cfg, energy_model=None)
    key_shapes = [(k, list(v.shape)) for k, v in m.state_dict().items()]
    w = gl.load_weights(name, key_shapes)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    return kind, cfg, m


out = {}
for name in ("small_vp_noise", "small_vp_score", "small_vp_velocity", "small_energy", "qgpo_noise"):
    kind, cfg, m = build(name)
    gen = torch.Generator().manual_seed(gl._seed_for(name, 77))
    x0 = torch.randn(5, cfg.x_size, generator=gen)
    cond = torch.randn(5, 17, generator=gen)
    out[name + "/x0"], out[name + "/cond"] = x0.numpy(), cond.numpy()
    for c in CASES:
        sc = EasyDict(type="DPMSolver", args=dict(order=c[0], device="cpu", steps=c[1], type=c[2], method=c[3],
                                                    solver_type=c[4], skip_type=c[5], denoise=c[6]))
        scales = (0.0, 1.0) if kind == "energy" else (None,)
        for gs in scales:
            kw = dict(t_span=None, x_0=x0, condition=cond, solver_config=sc)
            if gs is not None:
                kw["guidance_scale"] = gs
            try:
                traj = m.sample_forward_process(**kw)
            except Exception as e:  # noqa: BLE001 -- record reference failures instead of hiding them
                print("reference raises for", name, case_key(c), gs, type(e).__name__, e)
                RAISES.append("%s/%s: %s" % (name, case_key(c), type(e).__name__))
                continue
            key = "%s/%s%s" % (name, case_key(c), "" if gs is None else "_g%g" % gs)
            out[key] = traj.detach().numpy()
            print(key, tuple(traj.shape), float(traj[-1].abs().max()))
out["raises"] = np.array(sorted(set(RAISES)))
np.savez_compressed(os.path.join(HERE, "dpm.npz"), **out)
print("wrote dpm.npz", sum(a.nbytes for a in out.values()) // 1024, "KiB")
