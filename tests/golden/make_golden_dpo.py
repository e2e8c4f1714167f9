"""Golden vector for DiffusionModel.dpo_loss (grl/generative_models/diffusion_model/diffusion_model.py:1191-1233), from
the REFERENCE: loss and every parameter gradient of an unconditional D = 2 model against a second, frozen model.
Container-only script:   python tests/golden/make_golden_dpo.py   ->  tests/golden/dpo.npz
The draws (randn_like(x_w) first, then rand(B) for t) are captured by replaying the generator."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import golden_lib as gl  # noqa: E402
from ref_shims import EasyDict  # noqa: E402
from grl.generative_models.diffusion_model.diffusion_model import DiffusionModel  # noqa: E402

torch.set_num_threads(8)
cfg = EasyDict(gl.model_zoo()["swiss_noise"][1])
m, ref = DiffusionModel(cfg), DiffusionModel(cfg)
key_shapes = [(k, list(v.shape)) for k, v in m.state_dict().items()]
w = gl.load_weights("swiss_noise", key_shapes)
m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
# the frozen model: the same weights, perturbed deterministically
gen = torch.Generator().manual_seed(4242)
ref.load_state_dict({k: torch.from_numpy(v) + 0.02 * torch.randn(v.shape, generator=gen) * (0.0 if k.endswith("t_encoder.W") else 1.0)
                     for k, v in w.items()})
B = 48
data = torch.randn(B, 4, generator=gen)
seed = 9191
torch.manual_seed(seed)
noise = torch.randn(B, 2)
t = torch.rand(B) * (1.0 - 1e-5) + 1e-5
out = {"data": data.numpy(), "noise": noise.numpy(), "t": t.numpy(), "beta": np.float32(0.5)}
# (the frozen model's weights are not stored: tests rebuild them with the same generator calls, in the order of
# gl.load_weights' keys; a checksum guards that)
out["ref_checksum"] = np.float64(sum(float(v.double().sum()) for v in ref.state_dict().values()))
torch.manual_seed(seed)
loss = m.dpo_loss(ref, data, 0.5)
loss.backward()
out["loss"] = loss.detach().numpy()
for k, p in m.named_parameters():
    if p.grad is not None:
        flat = p.grad.detach().numpy().astype(np.float32).reshape(-1)
        out["gradsum/" + k] = np.asarray([flat.astype(np.float64).sum(), np.sqrt((flat.astype(np.float64) ** 2).sum())])
        out["gradsub/" + k] = flat[:: max(1, flat.size // 509)][:509].copy()
np.savez_compressed(os.path.join(HERE, "dpo.npz"), **out)
print("loss", float(loss), "wrote dpo.npz", sum(a.nbytes for a in out.values()) // 1024, "KiB")
