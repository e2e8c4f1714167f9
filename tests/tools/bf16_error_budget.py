#!/usr/bin/env python
"""CPU-only attribution of the bf16 integrator's error: which rounding owns the budget?

k_sample_bf16 keeps the state, the schedule and every accumulation in fp32 and rounds to bf16 (a) the state x when it
enters the net (dense1 / modify_x of the first block), (b) the time embedding and the condition / its embedding,
(c) the weights, (d) every activation that is the input of the next linear layer (and the identity residuals, which
are those same stored values).  This script re-runs the oracle's Euler chain on the golden qgpo_noise cases with each
of those roundings switched on alone, all together, and with the x_lo / cond_lo split-compensation of section 4.2 of
DESIGN.md, and prints max-abs errors of the final actions against the fp32 oracle.

    python tests/tools/bf16_error_budget.py > profiles/r03_bf16_error_budget.md

Test infrastructure: imports the oracle, never the product package."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import golden_lib as gl   # noqa: E402
import golden_util as gu  # noqa: E402
from oracle import grl_oracle as O  # noqa: E402


def rb(v):
    return v.bfloat16().float()


class Emu:
    """resnet_forward of the oracle with selectable bf16 roundings.  flags: subset of {x, tc, w, act}; split: subset of
    {x, c} -- those inputs enter as hi + lo bf16 pairs (error 2^-17 instead of 2^-9)."""

    def __init__(self, sd, flags, split=()):
        self.f, self.split = set(flags), set(split)
        self.sd = {k: (rb(v) if ("w" in self.f and k.endswith(".weight")) else v) for k, v in sd.items()}

    def inp(self, v, kind):
        if kind not in self.f:
            return v
        if kind in self.split or (kind == "tc" and "c" in self.split and False):
            hi = rb(v)
            return hi + rb(v - hi)
        return rb(v)

    def lin(self, key, v):
        return F.linear(v, self.sd[key + ".weight"], self.sd[key + ".bias"])

    def act(self, v):
        return rb(v) if "act" in self.f else v

    def block(self, pre, ste, xin):
        h1 = self.act(F.silu(self.lin(pre + "dense1.0", xin)) + self.lin(pre + "time_mlp.1", ste))
        h2 = F.silu(self.lin(pre + "dense2.0", h1))
        if (pre + "modify_x.weight") in self.sd:
            return self.act(h2 + self.lin(pre + "modify_x", xin))
        return self.act(h2 + xin)

    def net(self, t, x, cond):
        sd, pre = self.sd, "model.model.backbone."
        te = self.inp(O.fourier_embed(t, sd["model.model.t_encoder.W"]), "tc")
        if "c" in self.split and "tc" in self.f:
            hi = rb(cond)
            c = hi + rb(cond - hi)
        else:
            c = self.inp(cond, "tc")
        ce = self.act(F.silu(self.lin(pre + "pre_sort_condition.0", c)))
        z = self.act(F.silu(self.lin(pre + "sort_t.0", torch.cat([te, ce], -1))))
        ste = self.act(F.silu(self.lin(pre + "sort_t.2", z)))
        xin = self.inp(x, "x")
        d0 = self.block(pre + "first_block.", ste, xin)
        d1 = self.block(pre + "down_block.0.", ste, d0)
        d2 = self.block(pre + "down_block.1.", ste, d1)
        u = self.block(pre + "middle_block.", ste, d2)
        u = self.block(pre + "up_block.0.", ste, torch.cat([u, d2], -1))
        u = self.block(pre + "up_block.1.", ste, torch.cat([u, d1], -1))
        return self.lin(pre + "last_block", torch.cat([u, d0], -1))


def chain(emu, path, x0, cond, t_span):
    """Unguided noise-head reverse ODE (diffusion_process.py:879-887), Euler on the grid (oracle euler_ode)."""
    def drift(t, x):
        return O.reverse_ode_drift("noise_function", path, lambda a, b: emu.net(a, b, cond), t, x)
    return O.euler_ode(drift, x0, t_span)[-1]


def main():
    torch.manual_seed(0)
    sd = gu.weights("qgpo_noise")
    kind, cfg = gl.model_zoo()["qgpo_noise"]
    path = O.Path(cfg["path"])
    g = gu.load("sample_qgpo_noise.npz")
    rs = np.random.RandomState(5)
    big_x, big_c = torch.from_numpy(rs.randn(512, 6).astype(np.float32)), torch.from_numpy(rs.randn(512, 17).astype(np.float32))
    cases = [("golden T5 (3 rows: the 1.99e-2 case)", torch.from_numpy(g["xT_T5_bNone_x0"]), torch.from_numpy(g["cond"]), 5),
             ("golden T101 (3 rows)", torch.from_numpy(g["xT_T101_bNone_x0"]), torch.from_numpy(g["cond"]), 101),
             ("512 random rows, T5", big_x, big_c, 5),
             ("512 random rows, T11", big_x, big_c, 11),
             ("512 random rows, T101", big_x, big_c, 101)]
    variants = [("x only", ("x",), ()), ("t_emb + condition only", ("tc",), ()), ("weights only", ("w",), ()),
                ("inter-layer activations only", ("act",), ()), ("everything (the kernel today)", ("x", "tc", "w", "act"), ()),
                ("everything, x as hi + lo", ("x", "tc", "w", "act"), ("x",)),
                ("everything, x and condition as hi + lo", ("x", "tc", "w", "act"), ("x", "c"))]
    print("# r03: bf16 error budget of the fused integrator (CPU emulation, qgpo_noise trained weights)\n")
    print("Max-abs error of the final actions against the fp32 oracle; each column rounds ONE class of values to bf16.")
    print("Generated by `tests/tools/bf16_error_budget.py`.\n")
    print("| case | " + " | ".join(v[0] for v in variants) + " |")
    print("|---|" + "---|" * len(variants))
    with torch.no_grad():
        for name, x0, cond, T in cases:
            t_span = torch.linspace(0.0, 1.0, T)
            ref = chain(Emu(sd, ()), path, x0, cond, t_span)
            row = []
            for _, flags, split in variants:
                got = chain(Emu(sd, flags, split), path, x0, cond, t_span)
                row.append("%.2e" % float((got - ref).abs().max()))
            print("| %s (max abs action %.2f) | " % (name, float(ref.abs().max())) + " | ".join(row) + " |")


if __name__ == "__main__":
    main()
