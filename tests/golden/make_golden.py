"""Generate the golden fixtures in this directory by running the REFERENCE implementation
(/root/reference, imported read-only) on seeded inputs.  Run once in the build container:

    python tests/golden/make_golden.py

Only the resulting ``*.npz`` / ``manifest.json`` travel (the reference cannot).  Nothing in
``tests/``, ``bench.py`` or the package imports this file.

Third-party packages the reference needs but the container lacks are replaced by minimal
stand-ins that are only ever reached on code paths we do not exercise (``easydict``,
``tensordict``, ``treetensor``, ``ot``, ``wandb``, maxvit) -- plus the three integrators
(``torchdyn.numerics.odeint``, ``torchdiffeq.odeint``, ``torchsde.sdeint``), whose stepping
semantics are therefore NOT pinned by the reference: the stand-ins below
implement the documented fixed-grid explicit Euler / dt-driven Euler-Maruyama behaviour and
are the *specification* of this project's integrators.  Everything else in a golden (net,
Fourier embedding, schedules, drift closures, head conversions, guidance gradient, losses,
gradients, batch-shape bookkeeping) is computed by the reference's own code.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import golden_lib as gl  # noqa: E402


from ref_shims import EasyDict  # noqa: E402,F401  (installs the stand-ins, adds /root/reference to sys.path)
from grl.generative_models.diffusion_model.diffusion_model import DiffusionModel  # noqa: E402
from grl.generative_models.diffusion_model.energy_conditional_diffusion_model import (  # noqa: E402
    EnergyConditionalDiffusionModel,
)
from grl.generative_models.conditional_flow_model.independent_conditional_flow_model import (  # noqa: E402
    IndependentConditionalFlowModel,
)
from grl.numerical_methods.monte_carlo import MonteCarloSampler  # noqa: E402

torch.set_num_threads(8)
torch.set_grad_enabled(True)


def build(name):
    kind, cfg = gl.model_zoo()[name]
    cfg = EasyDict(cfg)
    if kind == "diffusion":
        m = DiffusionModel(cfg)
    elif kind == "energy":
        m = EnergyConditionalDiffusionModel(cfg, energy_model=None)
    else:
        m = IndependentConditionalFlowModel(cfg)
    key_shapes = [(k, list(v.shape)) for k, v in m.state_dict().items()]
    w = gl.load_weights(name, key_shapes)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    return kind, cfg, m, key_shapes


def T(a):
    return None if a is None else torch.from_numpy(np.asarray(a))


def N(t):
    return t.detach().cpu().numpy()


def grad_summary(model, full):
    out = {}
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        g = N(p.grad).astype(np.float32)
        if full:
            out["grad/" + k] = g
        else:
            flat = g.reshape(-1)
            out["gradsum/" + k] = np.asarray([flat.astype(np.float64).sum(),
                                              np.sqrt((flat.astype(np.float64) ** 2).sum())])
            out["gradsub/" + k] = flat[:: max(1, flat.size // 509)][:509].copy()
    return out


def zero_grad(model):
    for p in model.parameters():
        p.grad = None


manifest = {"torch": torch.__version__, "numpy": np.__version__, "models": {}, "files": []}


def save(fname, **arrs):
    arrs = {k: np.asarray(v) for k, v in arrs.items() if v is not None}
    np.savez_compressed(os.path.join(HERE, fname), **arrs)
    manifest["files"].append(fname)
    print("wrote", fname, sum(a.nbytes for a in arrs.values()) // 1024, "KiB")


HEADS = ("score_function", "noise_function", "velocity_function", "data_prediction_function")


def one_model(name, rows, full_grads):
    kind, cfg, m, key_shapes = build(name)
    manifest["models"][name] = dict(kind=kind, key_shapes=key_shapes)
    D = cfg.x_size
    S = cfg.model.args.backbone.args.get("condition_dim", None)
    inp = gl.make_inputs(name, rows, D, S)
    x, t, eps, cond = T(inp["x"]), T(inp["t"]), T(inp["eps"]), T(inp.get("cond"))
    out = dict(x=inp["x"], t=inp["t"], eps=inp["eps"], eps2=inp["eps2"], cond=inp.get("cond"))

    # (1) raw net
    with torch.no_grad():
        out["net"] = N(m.model(t, x, cond))

    if kind in ("diffusion", "energy"):
        with torch.no_grad():
            # (2) the four head conversions + reverse ODE / SDE closures
            # a head conversion the reference itself cannot evaluate (e.g. the as-written
            # data_prediction -> velocity formula broadcasts (B,) against (B,D)) is recorded
            # in the manifest as "raises" instead of as a vector
            for hname in ("score", "noise", "velocity", "data"):
                fn = getattr(m, {"score": "score_function", "noise": "noise_function",
                                 "velocity": "velocity_function", "data": "data_prediction_function"}[hname])
                try:
                    out["head_" + hname] = N(fn(t, x, cond))
                except (RuntimeError, NotImplementedError) as e:
                    manifest["models"][name].setdefault("raises", {})["head_" + hname] = type(e).__name__
            tt = 1.0 - t  # solver time; the closures evaluate the model at T - tt == t (up to rounding)
            out["t_solver"] = N(tt)

            def attempt(key, fn):
                try:
                    out[key] = N(fn())
                except (RuntimeError, NotImplementedError, TypeError) as e:
                    manifest["models"][name].setdefault("raises", {})[key] = type(e).__name__

            attempt("ode_drift", lambda: m.diffusion_process.reverse_ode(
                function=m.model, function_type=m.model_type, condition=cond).drift(tt, x))

            def _sde():
                return m.diffusion_process.reverse_sde(
                    function=m.model, function_type=m.model_type, condition=cond,
                    reverse_diffusion_function=m.reverse_diffusion_process.diffusion,
                    reverse_diffusion_squared_function=m.reverse_diffusion_process.diffusion_squared)
            attempt("sde_drift", lambda: _sde().drift(tt, x))
            attempt("sde_diffusion", lambda: _sde().diffusion(tt, x))
        # (5) losses + gradients, explicit (t, eps) captured by replaying the RNG draw order
        def loss_block(prefix, seed, call, call_rows=None):
            torch.manual_seed(seed)
            t_r = torch.rand(rows) * (1.0 - 1e-5) + 1e-5
            e_r = torch.randn(rows, D)
            zero_grad(m)
            torch.manual_seed(seed)
            try:
                loss = call()
                loss.backward()
            except (RuntimeError, NotImplementedError, TypeError) as e:
                manifest["models"][name].setdefault("raises", {})[prefix + "_loss"] = type(e).__name__
                return
            out[prefix + "_t"], out[prefix + "_eps"], out[prefix + "_loss"] = N(t_r), N(e_r), N(loss)
            out.update({prefix + "_" + k: v for k, v in grad_summary(m, full_grads).items()})
            if call_rows is not None:
                torch.manual_seed(seed)
                out[prefix + "_loss_rows"] = N(call_rows())

        loss_block("sm", 1234, lambda: m.score_matching_loss(x, cond, weighting_scheme="vanilla"),
                   lambda: m.score_matching_loss(x, cond, weighting_scheme="vanilla", average=False))
        loss_block("fm", 4321, lambda: m.flow_matching_loss(x, cond),
                   lambda: m.flow_matching_loss(x, cond, average=False))
    else:
        # ICFM: x0 = noise sample, x1 = data; t ~ rand * t_max, then randn_like(x0) inside direct_sample
        x0, x1 = eps, x
        torch.manual_seed(4321)
        t_r = torch.rand(rows) * 1.0
        e_r = torch.randn(rows, D)
        zero_grad(m)
        torch.manual_seed(4321)
        loss = m.flow_matching_loss(x0=x0, x1=x1, condition=cond)
        loss.backward()
        out["fm_t"], out["fm_eps"], out["fm_loss"] = N(t_r), N(e_r), N(loss)
        out.update({"fm_" + k: v for k, v in grad_summary(m, full_grads).items()})
        torch.manual_seed(4321)
        out["fm_loss_rows"] = N(m.flow_matching_loss(x0=x0, x1=x1, condition=cond, average=False))

    if kind == "energy":
        # (4) guidance gradient and guided score
        for s in (0.0, 1.0, 5.0):
            g = m.energy_guidance.calculate_energy_guidance(t, x.clone(), cond, guidance_scale=s)
            out["guidance_%g" % s] = N(g)
            out["guided_score_%g" % s] = N(m.score_function_with_energy_guidance(t, x.clone(), cond, s))
        with torch.no_grad():
            out["energy"] = N(m.energy_guidance(t, x, cond))

    save("model_%s.npz" % name, **out)

    # (6) sampling: trajectories for the four (batch_size, x_0) combinations
    NB, B = 3, 2
    sm = dict()
    x0s, conds = T(inp["x"][:NB]), (T(inp["cond"][:NB]) if S else None)
    for Tn in (5, 32, 101):
        if rows > 64 and Tn == 101 and D > 8:
            continue
        t_span = torch.linspace(0.0, 1.0, Tn)
        for bs, use_x0 in ((None, False), (B, False), (None, True), (B, True)):
            tag = "T%d_b%s_x%d" % (Tn, bs, int(use_x0))
            kw = dict(t_span=t_span, batch_size=bs, x_0=x0s if use_x0 else None)
            if S:
                kw["condition"] = conds
            n_rows = (bs or 1) * (NB if (use_x0 or S) else 1)
            torch.manual_seed(99)
            xT = torch.randn(n_rows, D)  # what gaussian_generator draws when x_0 is None
            torch.manual_seed(99)
            if kind == "energy":
                for gs in (0.0, 1.0):
                    torch.manual_seed(99)
                    traj = m.sample_forward_process(guidance_scale=gs, **kw)
                    sm["traj_%s_g%g" % (tag, gs)] = N(traj)
            else:
                try:
                    traj = m.sample_forward_process(**kw)
                except (RuntimeError, NotImplementedError, TypeError) as e:
                    manifest["models"][name].setdefault("raises", {})["traj_" + tag] = type(e).__name__
                    continue
                sm["traj_" + tag] = N(traj)
            if not use_x0:
                sm["xT_" + tag] = N(xT)
    sm["x0"] = N(x0s)
    if S:
        sm["cond"] = N(conds)
    # (6b) Euler-Maruyama with a caller-fixed Brownian increment buffer
    if kind in ("diffusion", "energy"):
        for Tn, dt in ((11, 0.1), (21, 0.05)):
            t_span = torch.linspace(0.0, 1.0, Tn)
            n_rows = NB * B if S else B
            rs = np.random.RandomState(5 + Tn)
            steps = Tn - 1
            dW = (rs.randn(steps, n_rows, D) * np.sqrt(dt)).astype(np.float32)

            class BM:
                def __init__(self):
                    self.k = 0

                def __call__(self, ta, tb):
                    w = torch.from_numpy(dW[self.k])
                    self.k += 1
                    return w

            solver_cfg = EasyDict(type="SDESolver", args=dict(library="torchsde", dt=dt, bm=BM()))
            kw = dict(t_span=t_span, batch_size=B, solver_config=solver_cfg)
            torch.manual_seed(77)
            xT = torch.randn(n_rows, D)
            if S:
                kw["condition"] = conds
            if kind == "energy":
                for gs in (0.0, 1.0):
                    solver_cfg.args["bm"] = BM()
                    torch.manual_seed(77)
                    sm["sde_T%d_g%g" % (Tn, gs)] = N(m.sample_forward_process(guidance_scale=gs, **kw))
            elif not S:
                # plain DiffusionModel + SDESolver only works unconditionally
                torch.manual_seed(77)
                try:
                    sm["sde_T%d" % Tn] = N(m.sample_forward_process(**kw))
                except (RuntimeError, NotImplementedError, TypeError) as e:
                    manifest["models"][name].setdefault("raises", {})["sde_T%d" % Tn] = type(e).__name__
                    continue
            else:
                continue
            sm["sde_T%d_dW" % Tn] = dW
            sm["sde_T%d_xT" % Tn] = N(xT)
    save("sample_%s.npz" % name, **sm)


# canonical nets: summarised gradients; small nets: every gradient element
one_model("qgpo_noise", 40, full_grads=False)
one_model("swiss_noise", 40, full_grads=False)
one_model("icfm_cond", 40, full_grads=False)
one_model("gvp_wide", 24, full_grads=False)
one_model("hopper_noise", 40, full_grads=False)
one_model("wideobs_noise", 40, full_grads=False)
for nm in gl.model_zoo():
    if nm.startswith("small_"):
        one_model(nm, 37, full_grads=True)

# (8) host-side rejection sampler used by the default score-matching weighting
from grl.numerical_methods.probability_path import GaussianConditionalProbabilityPath  # noqa: E402

path = GaussianConditionalProbabilityPath(EasyDict(gl.VP))
mc = MonteCarloSampler(unnormalized_pdf=lambda t: path.diffusion_squared(t) / (path.covariance(t) + 1e-6),
                       x_min=0.0, x_max=1.0)
torch.manual_seed(2024)
save("monte_carlo.npz", t=N(mc.sample(4096)), seed=2024)

# (9) the scalar schedules on a grid, incl. the as-written quirks
tg = torch.linspace(0.0, 1.0, 257)
sch = {}
for pname, pcfg in (("vp", gl.VP), ("gvp", gl.GVP)):
    p = GaussianConditionalProbabilityPath(EasyDict(pcfg))
    for fn in ("drift_coefficient", "diffusion", "diffusion_squared", "scale", "d_scale_dt", "std", "d_std_dt",
               "covariance"):
        sch[pname + "_" + fn] = N(getattr(p, fn)(tg))
sch["t"] = N(tg)
save("schedules.npz", **sch)

with open(os.path.join(HERE, "manifest.json"), "w") as f:
    json.dump(manifest, f, indent=1)
print("done")
