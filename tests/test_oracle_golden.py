"""The CPU oracle (oracle/grl_oracle.py) against the golden vectors captured from the reference's
own classes (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import grl_oracle as O

ALL = gu.CANON + gu.SMALL


def close(a, b, rtol=2e-5, atol=2e-6, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1.0, float(np.max(np.abs(b)))) if b.size else 1.0
    err = float(np.max(np.abs(a - b))) if b.size else 0.0
    assert err <= atol * scale + rtol * scale, "%s: max err %.3e (scale %.3e)" % (what, err, scale)


@pytest.mark.parametrize("name", ALL)
def test_net_heads_drifts(name):
    kind, cfg, info = gu.model_info(name)
    sd = gu.weights(name)
    g = gu.load("model_%s.npz" % name)
    x, t, cond = gu.T(g["x"]), gu.T(g["t"]), gu.T(g.get("cond"))
    net = O.intrinsic_forward(sd, t, x, cond)
    close(net, g["net"], what="net")
    if kind == "icfm":
        return
    path, mt = O.Path(cfg["path"]), cfg["model"]["type"]
    for h, fn in (("score", O.score_head), ("noise", O.noise_head), ("velocity", O.velocity_head),
                  ("data", O.data_head)):
        if "head_" + h in g:
            close(fn(mt, path, net, t, x), g["head_" + h], what="head_" + h, rtol=1e-4)
        else:
            assert info["raises"]["head_" + h]
    tt = gu.T(g["t_solver"])
    fn = lambda tau, y: O.intrinsic_forward(sd, tau, y, cond)  # noqa: E731
    close(O.reverse_ode_drift(mt, path, fn, tt, x), g["ode_drift"], what="ode_drift", rtol=1e-4)
    rpath = O.Path(cfg["reverse_path"])
    close(O.reverse_sde_drift(mt, path, rpath, fn, tt, x), g["sde_drift"], what="sde_drift", rtol=1e-4)
    close(O.reverse_sde_diffusion(rpath, tt, x), g["sde_diffusion"], what="sde_diffusion")


@pytest.mark.parametrize("name", ["qgpo_noise", "small_energy"])
def test_energy_guidance(name):
    kind, cfg, info = gu.model_info(name)
    sd = gu.weights(name)
    g = gu.load("model_%s.npz" % name)
    x, t, cond = gu.T(g["x"]), gu.T(g["t"]), gu.T(g["cond"])
    close(O.intrinsic_forward(sd, t, x, cond, pre="energy_guidance.model.", backbone="ConcatenateMLP"),
          g["energy"], what="energy")
    path = O.Path(cfg["path"])
    for s in (0.0, 1.0, 5.0):
        close(O.energy_guidance(sd, t, x, cond, s), g["guidance_%g" % s], what="guidance")
        close(O.guided_score(sd, cfg["model"]["type"], path, t, x, cond, s), g["guided_score_%g" % s],
              what="guided", rtol=1e-4)


def _grads(sd, keys, loss):
    ps = [sd[k] for k in keys]
    return dict(zip(keys, torch.autograd.grad(loss, ps, allow_unused=True)))


@pytest.mark.parametrize("name", ALL)
def test_losses_and_grads(name):
    kind, cfg, info = gu.model_info(name)
    sd = gu.weights(name)
    g = gu.load("model_%s.npz" % name)
    x, cond = gu.T(g["x"]), gu.T(g.get("cond"))
    train_keys = [k for k in sd if k.startswith("model.") and not k.endswith("t_encoder.W")]
    for k in train_keys:
        sd[k].requires_grad_(True)
    fn = lambda t, y: O.intrinsic_forward(sd, t, y, cond)  # noqa: E731
    cases = []
    if kind == "icfm":
        sigma = cfg["path"]["sigma"]
        cases.append(("fm", lambda avg: O.icfm_flow_matching_loss(
            fn, gu.T(g["eps"]), x, gu.T(g["fm_t"]), gu.T(g["fm_eps"]), sigma, average=avg)))
    else:
        path, mt = O.Path(cfg["path"]), cfg["model"]["type"]
        if "sm_loss" in g:
            cases.append(("sm", lambda avg: O.score_matching_loss(
                mt, path, fn, x, gu.T(g["sm_t"]), gu.T(g["sm_eps"]), average=avg)))
        if "fm_loss" in g:
            cases.append(("fm", lambda avg: O.flow_matching_loss(
                mt, path, fn, x, gu.T(g["fm_t"]), gu.T(g["fm_eps"]), average=avg)))
    assert cases
    for pfx, call in cases:
        loss = call(True)
        assert abs(float(loss) - float(g[pfx + "_loss"])) <= 1e-5 * abs(float(g[pfx + "_loss"])) + 1e-7, pfx
        close(call(False).detach(), g[pfx + "_loss_rows"], what=pfx + "_rows", rtol=1e-4)
        grads = _grads(sd, train_keys, loss)
        n_checked = 0
        for k in train_keys:
            gr = grads[k]
            if pfx + "_grad/" + k in g:
                close(gr, g[pfx + "_grad/" + k], what=k, rtol=2e-4, atol=1e-6)
                n_checked += 1
            elif pfx + "_gradsub/" + k in g:
                flat = gr.reshape(-1)
                sub = flat[:: max(1, flat.numel() // 509)][:509]
                close(sub, g[pfx + "_gradsub/" + k], what=k, rtol=2e-4, atol=1e-6)
                s = g[pfx + "_gradsum/" + k]
                assert abs(float(flat.double().pow(2).sum().sqrt()) - s[1]) <= 2e-4 * s[1] + 1e-7
                n_checked += 1
        assert n_checked > 10


def _sample_cases(name):
    kind, cfg, info = gu.model_info(name)
    s = gu.load("sample_%s.npz" % name)
    S = gu.cond_dim(cfg)
    for key in sorted(s):
        if not key.startswith("traj_"):
            continue
        parts = key.split("_")  # traj_T5_bNone_x0[_g1]
        Tn = int(parts[1][1:])
        bs = None if parts[2] == "bNone" else int(parts[2][1:])
        use_x0 = parts[3] == "x1"
        gs = float(parts[4][1:]) if len(parts) > 4 else None
        yield key, Tn, bs, use_x0, gs, s, S


@pytest.mark.parametrize("name", ALL)
def test_sampling_trajectories(name):
    kind, cfg, info = gu.model_info(name)
    sd = gu.weights(name)
    n = 0
    for key, Tn, bs, use_x0, gs, s, S in _sample_cases(name):
        if Tn == 101 and name in gu.CANON:
            continue  # covered on the GPU path; keeps the CPU suite short
        t_span = torch.linspace(0.0, 1.0, Tn)
        x_0 = gu.T(s["x0"]) if use_x0 else None
        x_T = None if use_x0 else gu.T(s["xT_T%d_b%s_x0" % (Tn, bs)])
        out = O.sample_forward_process(sd, cfg, kind, t_span, batch_size=bs, x_0=x_0,
                                       condition=gu.T(s["cond"]) if S else None, x_T=x_T,
                                       guidance_scale=gs if gs is not None else 1.0)
        close(out, s[key], what=key, rtol=2e-4, atol=2e-5)
        n += 1
    assert n >= 4


@pytest.mark.parametrize("name", ["small_energy", "small_uncond_noise", "swiss_noise"])
def test_sde_sampling(name):
    kind, cfg, info = gu.model_info(name)
    sd = gu.weights(name)
    s = gu.load("sample_%s.npz" % name)
    S = gu.cond_dim(cfg)
    n = 0
    for Tn, dt in ((11, 0.1), (21, 0.05)):
        for key in [k for k in s if k.startswith("sde_T%d" % Tn) and not k.endswith(("dW", "xT"))]:
            gs = float(key.split("_g")[1]) if "_g" in key else 1.0
            out = O.sample_forward_process(
                sd, cfg, kind, torch.linspace(0.0, 1.0, Tn), batch_size=2,
                condition=gu.T(s["cond"]) if S else None, x_T=gu.T(s["sde_T%d_xT" % Tn]),
                guidance_scale=gs, solver="SDESolver", dt=dt, increments=gu.T(s["sde_T%d_dW" % Tn]))
            close(out, s[key], what=key, rtol=5e-4, atol=5e-5)
            n += 1
    assert n >= 2


def test_schedules():
    g = gu.load("schedules.npz")
    t = gu.T(g["t"])
    for pname, pcfg in (("vp", dict(type="linear_vp_sde", beta_0=0.1, beta_1=20.0)), ("gvp", dict(type="gvp"))):
        p = O.Path(pcfg)
        for gold, fn in (("drift_coefficient", p.f), ("diffusion", p.g), ("diffusion_squared", p.g2),
                         ("scale", p.scale), ("d_scale_dt", p.d_scale), ("std", p.std), ("d_std_dt", p.d_std),
                         ("covariance", p.cov)):
            a, b = fn(t).numpy(), g[pname + "_" + gold]
            fin = np.isfinite(b)
            assert np.array_equal(np.isfinite(a), fin), gold
            np.testing.assert_allclose(a[fin], b[fin], rtol=1e-6, atol=1e-7, err_msg=pname + gold)


FIXED_X_MODELS = ("small_vp_noise", "small_vp_score", "small_gvp_velocity", "small_uncond_noise", "small_energy", "qgpo_noise")


@pytest.mark.parametrize("name", FIXED_X_MODELS)
def test_sampling_with_fixed_x(name):
    """sample_forward_process_with_fixed_x of the reference (tests/golden/make_golden_fixed_x.py): pinned, free and
    fractionally masked components; x_0 given / drawn; extra batch sizes; guided and unguided; ODE and (where the
    reference itself runs) Euler-Maruyama."""
    kind, cfg, info = gu.model_info(name)
    sd = gu.weights(name)
    g = gu.load("fixed_x.npz")
    pre = name + "/"
    x0, fx, mask = gu.T(g[pre + "x0"]), gu.T(g[pre + "fixed_x"]), gu.T(g[pre + "mask"])
    cond = gu.T(g[pre + "cond"]) if pre + "cond" in g else None
    t_span = torch.linspace(0.0, 1.0, 7)
    rel = name in ("small_vp_score",)
    n = 0
    for tag, B, use_x0 in (("x0", None, True), ("B2_x0", 2, True), ("B3_draw", 3, False)):
        for gs in ((0.0, 1.0) if kind == "energy" else (None,)):
            key = "%sode_%s%s" % (pre, tag, "" if gs is None else "_g%g" % gs)
            traj = O.sample_forward_process_with_fixed_x(
                sd, cfg, kind, t_span, fx, mask, gu.T(g["%sode_%s_noise" % (pre, tag)]), batch_size=B,
                x_0=x0 if use_x0 else None, condition=cond, x_T=None if use_x0 else gu.T(g["%sode_%s_xT" % (pre, tag)]),
                guidance_scale=1.0 if gs is None else gs)
            close(traj, g[key], what=key, rtol=1e-4 if rel else 2e-5)
            # pinned components never move, free rows follow the ordinary sampler's arithmetic
            n += 1
    for gs in ((0.0, 1.0) if kind == "energy" else (None,)):
        key = "%ssde%s" % (pre, "" if gs is None else "_g%g" % gs)
        if key not in g:
            assert any(r.startswith(key[:-3] if gs is not None else key) or r.startswith(pre + "sde") for r in g["raises"])
            continue
        traj = O.sample_forward_process_with_fixed_x(
            sd, cfg, kind, torch.linspace(0.0, 1.0, 5), fx, mask, gu.T(g[pre + "sde_noise"]), x_0=x0, condition=cond,
            guidance_scale=1.0 if gs is None else gs, solver="SDESolver", dt=0.25, increments=gu.T(g[pre + "sde_dW"]))
        close(traj, g[key], what=key, rtol=1e-4)
        n += 1
    assert n >= 3


FORWARD_MODELS = ("small_vp_noise", "small_vp_score", "small_vp_velocity", "small_vp_data", "small_gvp_velocity",
                  "small_uncond_noise", "small_icfm", "swiss_noise", "icfm_cond")


@pytest.mark.parametrize("name", FORWARD_MODELS)
def test_forward_sample_process(name):
    """forward_sample_process of the reference (tests/golden/make_golden_forward.py): the data -> latent direction of
    every head on both paths, the ICFM's reversed flow, and the forward SDE where the reference itself runs."""
    kind, cfg, info = gu.model_info(name)
    sd = gu.weights(name)
    g = gu.load("forward.npz")
    pre = name + "/"
    x = gu.T(g[pre + "x"])
    cond = gu.T(g[pre + "cond"]) if pre + "cond" in g else None
    traj = O.forward_sample_process(sd, cfg, kind, x, torch.linspace(0.05, 1.0, 8), cond)
    close(traj, g[pre + "ode"], what=pre + "ode", rtol=1e-4)
    if pre + "sde" in g:
        traj = O.forward_sample_process(sd, cfg, kind, x, torch.tensor([0.05, 0.3, 0.55, 0.8, 1.0]), cond,
                                        solver="SDESolver", dt=0.25, increments=gu.T(g[pre + "sde_dW"]))
        close(traj, g[pre + "sde"], what=pre + "sde", rtol=1e-4)
    elif kind == "diffusion":
        assert any(r.startswith(pre + "sde") for r in g["raises"])


@pytest.mark.parametrize("name", ["small_icfm", "icfm_cond"])
def test_icfm_masked_loss_and_sampling(name):
    """flow_matching_loss_with_mask (+ its element-wise loss) and sample_with_mask of the reference
    (tests/golden/make_golden_icfm_mask.py)."""
    kind, cfg, info = gu.model_info(name)
    sd = gu.weights(name)
    g = gu.load("icfm_mask.npz")
    pre = name + "/"
    x0, x1, cond, mask = (gu.T(g[pre + k]) for k in ("x0", "x1", "cond", "mask"))
    fn = lambda t, y: O.intrinsic_forward(sd, t, y, cond)  # noqa: E731
    sigma = cfg["path"]["sigma"]
    loss, el = O.icfm_flow_matching_loss_with_mask(fn, x0, x1, gu.T(g[pre + "t"]), gu.T(g[pre + "eps"]), sigma, mask)
    close(loss, g[pre + "loss"], what="masked loss")
    close(el, g[pre + "loss_elements"], what="element-wise loss")
    rows, _ = O.icfm_flow_matching_loss_with_mask(fn, x0, x1, gu.T(g[pre + "t"]), gu.T(g[pre + "eps"]), sigma, mask, average=False)
    close(rows, g[pre + "loss_rows"], what="masked loss rows")
    traj = O.sample_forward_process(sd, cfg, "icfm", torch.linspace(0.0, 1.0, 6), x_0=x0, condition=cond)
    close(torch.where(mask, x1, traj), g[pre + "sample_traj"], what="sample_with_mask")


LOGPROB_MODELS = ("small_vp_noise", "small_vp_score", "small_gvp_velocity", "small_uncond_noise", "small_icfm")


@pytest.mark.parametrize("name", LOGPROB_MODELS)
def test_log_likelihood(name):
    """compute_likelihood / log_prob of the reference (tests/golden/make_golden_logprob.py): exact trace and
    Hutchinson-Skilling estimator with replayed probes."""
    kind, cfg, info = gu.model_info(name)
    sd = gu.weights(name)
    g = gu.load("logprob.npz")
    pre = name + "/"
    x, grid = gu.T(g[pre + "x"]), gu.T(g[pre + "grid"])
    cond = gu.T(g[pre + "cond"]) if pre + "cond" in g else None
    close(O.log_likelihood(sd, cfg, kind, x, grid, cond, hutchinson=False), g[pre + "exact"], what="exact", rtol=2e-4)
    close(O.log_likelihood(sd, cfg, kind, x, grid, cond, probes=gu.T(g[pre + "probes"])), g[pre + "hutchinson"],
          what="hutchinson", rtol=2e-4)
