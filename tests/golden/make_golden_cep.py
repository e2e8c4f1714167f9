"""Golden vectors for the CEP energy_guidance_loss, produced by the REFERENCE's
EnergyConditionalDiffusionModel.energy_guidance_loss + autograd on the CPU.  Container-only script:
    python tests/golden/make_golden_cep.py        ->  tests/golden/cep.npz
The reference draws t (torch.rand) and then the noise (torch.randn_like in direct_sample) from the global
generator; the script reseeds and repeats the two draws in that order to record them.  The energy model is a
small fixed function defined here (the reference accepts any module); its outputs are stored, so the tests do
not need it.  (K = 1 raises in the reference: squeeze(dim=-1) drops the group axis before LogSoftmax(dim=1).)"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import golden_lib as gl  # noqa: E402
from ref_shims import EasyDict  # noqa: E402
from grl.generative_models.diffusion_model.energy_conditional_diffusion_model import (  # noqa: E402
    EnergyConditionalDiffusionModel,
)


class ToyEnergy(nn.Module):
    def forward(self, x, c=None):
        e = torch.sin(x).sum(-1, keepdim=True)
        return e if c is None else e + 0.3 * torch.tanh(c).sum(-1, keepdim=True)


out = {}
# (model, B, K, alpha)
for name, B, K, alpha in (("small_energy", 6, 4, 1.0), ("qgpo_noise", 5, 16, 3.0), ("qgpo_noise", 3, 2, 3.0)):
    kind, cfg = gl.model_zoo()[name]
    cfg = EasyDict(cfg)
    m = EnergyConditionalDiffusionModel(cfg, energy_model=ToyEnergy())
    m.alpha = alpha
    w = gl.load_weights(name, [(k, list(v.shape)) for k, v in m.state_dict().items()])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    for p in m.energy_guidance.parameters():     # also the Fourier frequencies (requires_grad=False by default)
        p.requires_grad_(True)
    gen = torch.Generator().manual_seed(gl._seed_for(name, 1000 + K))
    x = torch.randn(B, K, cfg.x_size, generator=gen)
    cond = torch.randn(B, 17, generator=gen)
    seed = 4242 + K
    torch.manual_seed(seed)
    loss = m.energy_guidance_loss(x, cond)
    loss.backward()
    torch.manual_seed(seed)
    t = torch.rand((B,)) * (1.0 - 1e-3) + 1e-3
    noise = torch.randn_like(x)
    key = "%s_B%d_K%d" % (name, B, K)
    with torch.no_grad():
        energy = ToyEnergy()(x.reshape(B * K, -1), torch.stack([cond] * K, dim=1).reshape(B * K, -1)).reshape(B, K)
    out[key + "/x"], out[key + "/cond"], out[key + "/t"], out[key + "/noise"] = x.numpy(), cond.numpy(), t.numpy(), noise.numpy()
    out[key + "/energy"], out[key + "/alpha"], out[key + "/loss"] = energy.numpy(), np.float32(alpha), loss.detach().numpy()
    for n, p in m.energy_guidance.named_parameters():
        out[key + "/grad/energy_guidance." + n] = p.grad.numpy()
    print(key, float(loss), {n: float(p.grad.abs().max()) for n, p in list(m.energy_guidance.named_parameters())[:3]})
np.savez_compressed(os.path.join(HERE, "cep.npz"), **out)
print("wrote cep.npz", sum(a.nbytes for a in out.values()) // 1024, "KiB")
