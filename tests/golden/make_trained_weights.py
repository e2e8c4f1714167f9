"""Train the VP-path golden models (golden_lib.TRAINED) with the REFERENCE's own loss and store the weights in
weights_<name>.npz.  Container-only (imports /root/reference through ref_shims); run before make_golden*.py:

    python tests/golden/make_trained_weights.py

Start: the seeded init of golden_lib.make_weights.  Then `steps` Adam steps of the reference's
``DiffusionModel.score_matching_loss(x, cond, weighting_scheme="vanilla")`` (``flow_matching_loss`` for the velocity
head) on golden_lib.toy_data -- actions in [-1, 1] -- so that 100-step samples end at |x| = O(1) and the parity tests
can assert north_star's ABSOLUTE tolerances on them (SURVEY.md section 8c recipe).  Energy-guidance nets keep their
seeded init (no reference-side training signal without a critic).  Training is not bit-reproducible across machines,
which is why the result is stored rather than regenerated."""
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

ONLY = sys.argv[1:]
for name in gl.TRAINED:
    if ONLY and name not in ONLY:
        continue
    kind, cfg = gl.model_zoo()[name]
    cfg = EasyDict(cfg)
    m = DiffusionModel(cfg) if kind == "diffusion" else EnergyConditionalDiffusionModel(cfg, energy_model=None)
    key_shapes = [(k, list(v.shape)) for k, v in m.state_dict().items()]
    m.load_state_dict({k: torch.from_numpy(v) for k, v in gl.make_weights(name, key_shapes).items()})
    D = cfg.x_size
    S = cfg.model.args.backbone.args.get("condition_dim", None)
    big = name in ("qgpo_noise", "swiss_noise", "hopper_noise", "wideobs_noise")
    steps, batch, lr = (400, 1024, 1e-3) if big else (600, 512, 3e-3)
    params = [p for k, p in m.named_parameters() if k.startswith("model.") and p.requires_grad]
    opt = torch.optim.Adam(params, lr=lr)
    torch.manual_seed(gl._seed_for(name + "/train", 0))
    head = cfg.model.type
    for it in range(steps):
        x, c = gl.toy_data(name, batch, D, S, it)
        x, c = torch.from_numpy(x), (torch.from_numpy(c) if c is not None else None)
        if head == "velocity_function":
            loss = m.flow_matching_loss(x, c)
        else:
            loss = m.score_matching_loss(x, c, weighting_scheme="vanilla")
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        if it % 100 == 0 or it == steps - 1:
            print(name, it, float(loss), flush=True)
    if name in gl.STORED_F16:
        with torch.no_grad():
            for v in m.state_dict().values():
                if v.dtype == torch.float32:
                    v.copy_(v.half().float())
    # how large do 100-step samples get?
    with torch.no_grad():
        xt, ct = gl.toy_data(name, 64, D, S, 10 ** 6)
        kw = dict(t_span=torch.linspace(0.0, 1.0, 101), x_0=torch.randn(64, D))
        if S:
            kw["condition"] = torch.from_numpy(ct)
        if kind == "energy":
            kw["guidance_scale"] = 1.0
        out = m.sample(**kw)
        print(name, "max |sample| after 100 Euler steps: %.3f" % float(out.abs().max()), flush=True)
    np.savez_compressed(os.path.join(HERE, "weights_%s.npz" % name),
                        **{k: v.detach().numpy().astype(np.float16 if name in gl.STORED_F16 else np.float32)
                           for k, v in m.state_dict().items()})
print("done")
