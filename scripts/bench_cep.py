#!/usr/bin/env python
"""Timing of the QGPO energy-guidance training step: CEP energy_guidance_loss
forward + backward (one grlhip_cep_loss_fwd_bwd call) + Adam, energy net ConcatenateMLP [55, 256, 256] -> 1,
B states x K = 16 support actions.  Beside it: the same step written in plain torch eager on the same GPU.
    python scripts/bench_cep.py [--states 4096] [--steps 50]"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
FLOP_PER_ROW = 6 * (55 * 256 + 256 * 256 + 256)   # fwd + bwd = 3 x 2 x MAC


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--states", type=int, default=4096)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--steps", type=int, default=50)
    args = ap.parse_args()
    DEV = "cuda:0"
    import golden_lib as gl
    import generativerl_preview_amd as G
    torch.manual_seed(0)
    kind, cfg = gl.model_zoo()["qgpo_noise"]
    m = G.EnergyConditionalDiffusionModel(dict(cfg, device=DEV)).to(DEV)
    m.alpha = 3.0
    B, K = args.states, args.k
    x = torch.randn(B, K, 6, device=DEV)
    cond = torch.randn(B, 17, device=DEV)
    energy = torch.randn(B, K, device=DEV)
    opt = torch.optim.Adam(m.energy_guidance.parameters(), lr=3e-4)

    def step():
        opt.zero_grad(set_to_none=False)
        loss = m.energy_guidance_loss(x, cond, energy=energy)
        loss.backward()
        opt.step()
        return loss

    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "t_encoder" not in k)
          for k, v in m.energy_guidance.state_dict().items()}
    pre = "model.model.backbone.model.model.linear%d."
    opt2 = torch.optim.Adam([v for v in sd.values() if v.requires_grad], lr=3e-4)

    def step_torch():
        opt2.zero_grad(set_to_none=False)
        t = torch.rand(B, device=DEV) * (1 - 1e-3) + 1e-3
        tr = t[:, None].expand(B, K)
        xt = m.path.scale(tr)[..., None] * x + m.path.std(tr)[..., None] * torch.randn_like(x)
        ph = tr.reshape(-1, 1) * sd["model.model.t_encoder.W"][None]
        h = torch.cat([torch.sin(ph), torch.cos(ph), xt.reshape(B * K, 6), cond[:, None].expand(B, K, 17).reshape(B * K, 17)], -1)
        for i in range(2):
            h = F.silu(F.linear(h, sd[(pre % i) + "weight"], sd[(pre % i) + "bias"]))
        f = F.linear(h, sd[(pre % 2) + "weight"], sd[(pre % 2) + "bias"]).reshape(B, K)
        loss = -(torch.softmax(energy * 3.0, 1) * torch.log_softmax(f, 1)).sum(-1).mean()
        loss.backward()
        opt2.step()
        return loss

    out = {}
    for name, fn in (("hip", step), ("torch_eager", step_torch)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = fn()
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / args.steps
        assert bool(torch.isfinite(loss))
    rows = B * K
    print(json.dumps({"config": "QGPO CEP energy_guidance_loss fwd+bwd+Adam, %d states x %d actions, energy net [55,256,256]->1" % (B, K),
                      "engine": "f32 batch engine (grlhip_cep_loss_fwd_bwd)", "ms_per_step": out["hip"] * 1e3,
                      "rows_per_s": rows / out["hip"], "tflops": rows * FLOP_PER_ROW / out["hip"] / 1e12,
                      "torch_eager_ms_per_step": out["torch_eager"] * 1e3}), flush=True)


if __name__ == "__main__":
    main()
