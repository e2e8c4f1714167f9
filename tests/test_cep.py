"""Contrastive energy prediction, EnergyConditionalDiffusionModel.energy_guidance_loss:
the oracle restatement against goldens produced by the reference's own loss + autograd
(tests/golden/make_golden_cep.py), and the package's loss -- energy-net forward, group softmax cross-entropy and
backward in one grlhip_cep_loss_fwd_bwd call -- against the same goldens and against the oracle on the GPU."""
import numpy as np
import pytest
import torch

import golden_lib as gl
import golden_util as gu
import util_models as um
from oracle import grl_oracle as O

CASES = ("small_energy_B6_K4", "qgpo_noise_B5_K16", "qgpo_noise_B3_K2")
PFX = "energy_guidance.model."


def rel_close(a, b, tol, what):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = float(np.max(np.abs(b)))
    err = float(np.max(np.abs(a - b)))
    assert err <= tol * scale + 1e-6, "%s: max-abs err %.3e (scale %.2e, tol %.1e)" % (what, err, scale, tol)


def _inputs(g, key):
    return {k: torch.from_numpy(np.asarray(g[key + "/" + k])) for k in ("x", "cond", "t", "noise", "energy")}, float(g[key + "/alpha"])


def _oracle(name, inp, alpha):
    sd = {k: (v.clone().requires_grad_(True) if k.startswith("energy_guidance.") else v) for k, v in gu.weights(name).items()}
    path = O.Path(gl.model_zoo()[name][1]["path"])
    loss = O.energy_guidance_loss(sd, path, inp["x"], inp["cond"], inp["energy"], inp["t"], inp["noise"], alpha)
    keys = [k for k in sd if k.startswith("energy_guidance.")]
    return loss.detach(), dict(zip(keys, torch.autograd.grad(loss, [sd[k] for k in keys])))


@pytest.mark.parametrize("key", CASES)
def test_oracle_cep_loss_matches_the_reference(key):
    g = gu.load("cep.npz")
    name = key.split("_B")[0]
    inp, alpha = _inputs(g, key)
    loss, grads = _oracle(name, inp, alpha)
    rel_close(loss, g[key + "/loss"], 1e-6, "loss")
    n = 0
    for k, v in grads.items():
        rel_close(v, g[key + "/grad/" + k], 1e-4, k)
        n += 1
    assert n == 7  # Fourier W + 3 x (weight, bias)


def _package_grads(m):
    return {"energy_guidance." + k: p.grad for k, p in m.energy_guidance.named_parameters()}


@pytest.mark.gpu
@pytest.mark.parametrize("key", CASES)
def test_gpu_cep_loss_and_grads_vs_golden(key):
    """loss within 1e-5 relative, every parameter gradient within 2e-4 of its own scale"""
    DEV = "cuda:0"
    g = gu.load("cep.npz")
    name = key.split("_B")[0]
    inp, alpha = _inputs(g, key)
    kind, cfg, m = um.build(name, DEV)
    m.alpha = alpha
    for p in m.energy_guidance.parameters():
        p.requires_grad_(True)
    loss = m.energy_guidance_loss(inp["x"].to(DEV), inp["cond"].to(DEV), t=inp["t"].to(DEV), noise=inp["noise"].to(DEV),
                                  energy=inp["energy"].to(DEV))
    loss.backward()
    rel_close(loss, g[key + "/loss"], 1e-5, "loss")
    grads = _package_grads(m)
    assert len(grads) == 7
    for k, v in grads.items():
        rel_close(v, g[key + "/grad/" + k], 2e-4, k)
    for k, p in m.model.named_parameters():
        assert p.grad is None, "the diffusion model takes no gradient from the CEP loss: " + k


@pytest.mark.gpu
def test_gpu_cep_loss_spans_chunks_uses_energy_model_and_draws():
    """80 000 rows (two passes of whole groups) with the energy model evaluated by the method itself; the internal
    draws follow the reference's order (t, then the noise)."""
    DEV = "cuda:0"
    name = "small_energy"
    kind, cfg, m = um.build(name, DEV)
    m.alpha = 2.0
    B, K = 5000, 16
    gen = torch.Generator().manual_seed(5)
    x, cond = torch.randn(B, K, 6, generator=gen), torch.randn(B, 17, generator=gen)

    class Energy(torch.nn.Module):
        def forward(self, a, s):
            return torch.cos(a).sum(-1, keepdim=True) - 0.1 * s.pow(2).sum(-1, keepdim=True)

    m.energy_model = Energy()
    torch.manual_seed(99)
    loss = m.energy_guidance_loss(x.to(DEV), cond.to(DEV))
    loss.backward()
    torch.manual_seed(99)
    t = torch.rand((B,), device=DEV) * (1.0 - 1e-3) + 1e-3
    noise = torch.randn(B, K, 6, device=DEV)
    energy = Energy()(x.reshape(B * K, 6), cond.unsqueeze(1).expand(B, K, 17).reshape(B * K, 17)).reshape(B, K)
    want, grads = _oracle(name, dict(x=x, cond=cond, energy=energy, t=t.cpu(), noise=noise.cpu()), 2.0)
    rel_close(loss, want, 1e-5, "loss")
    got = _package_grads(m)
    for k, v in grads.items():
        if k.endswith("t_encoder.W"):
            assert got[k] is None   # GaussianFourierProjectionTimeEncoder: requires_grad=False unless configured
            continue
        rel_close(got[k], v, 3e-4, k)


@pytest.mark.gpu
def test_gpu_cep_rejects_what_the_reference_rejects():
    DEV = "cuda:0"
    from generativerl_preview_amd import _lib as L
    kind, cfg, m = um.build("small_energy", DEV)
    with pytest.raises(IndexError):      # K = 1: the reference's squeeze drops the group axis, LogSoftmax(dim=1) raises
        m.energy_guidance_loss(torch.zeros(4, 1, 6, device=DEV), torch.zeros(4, 17, device=DEV), energy=torch.zeros(4, 1, device=DEV))
    with pytest.raises(NotImplementedError):
        m.energy_guidance_loss(dict(a=torch.zeros(4, 2, 6)), None)
    with pytest.raises(L.GrlHipError):   # no CPU fallback
        um.build("small_energy", "cpu")[2].energy_guidance_loss(torch.zeros(4, 2, 6), torch.zeros(4, 17), energy=torch.zeros(4, 2))
