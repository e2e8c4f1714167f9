#!/usr/bin/env python
"""Timing of the GMPG-style policy-gradient step: actions sampled with grad from the ICFM
flow policy ([512,256,128], T Euler steps), scored by a double-Q critic, loss = -mean(min Q), backward through critic
and Euler chain (grlhip_net_backward + grlhip_sample_ode_backward), Adam on the policy.
    python scripts/bench_with_grad.py [--rows 4096] [--t 32] [--steps 10]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
MAC_PER_ROW = 1626656


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4096)
    ap.add_argument("--t", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    DEV = "cuda:0"
    import golden_lib as gl
    import generativerl_preview_amd as G
    from generativerl_preview_amd.rl_modules.value_network import DoubleQNetwork
    torch.manual_seed(0)
    m = G.IndependentConditionalFlowModel(dict(gl.icfm_cfg(6, [512, 256, 128], 17), device=DEV)).to(DEV)
    q = DoubleQNetwork(dict(backbone=dict(type="ConcatenateMLP", args=dict(hidden_sizes=[23, 256, 256], output_size=1,
                                                                          activation="relu")))).to(DEV).requires_grad_(False)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    R, T = args.rows, args.t
    state = torch.randn(R, 17, device=DEV)
    t_span = torch.linspace(0, 1, T + 1, device=DEV)

    def step():
        opt.zero_grad(set_to_none=False)
        a = m.sample(t_span=t_span, condition=state, with_grad=True)
        loss = -q(a, state).mean()
        loss.backward()
        opt.step()
        return loss

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    assert bool(torch.isfinite(loss))
    flop = R * T * MAC_PER_ROW * 2 * 4   # forward + (recomputed forward + 2x backward) per Euler step
    print(json.dumps({"config": "GMPG-style policy-gradient step: ICFM [512,256,128] sample with_grad (%d Euler steps) -> double-Q -> backward + Adam, %d states" % (T, R),
                      "engine": "f32 batch engine (grlhip_sample_ode + grlhip_sample_ode_backward)", "ms_per_step": dt * 1e3,
                      "states_per_s": R / dt, "tflops": flop / dt / 1e12}), flush=True)


if __name__ == "__main__":
    main()
