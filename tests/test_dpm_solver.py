"""DPM-Solver: the oracle restatement against goldens produced by the reference's own
DPMSolver (tests/golden/make_golden_dpm.py), and the package's DPMSolver -- every model evaluation a HIP forward
through the C ABI -- against the same goldens on the GPU."""
import numpy as np
import pytest
import torch

import golden_lib as gl
import golden_util as gu
import util_models as um
from oracle import grl_oracle as O

MODELS = ("small_vp_noise", "small_vp_score", "small_vp_velocity", "small_energy", "qgpo_noise")


def _load():
    return gu.load("dpm.npz")


def _cases(g, name):
    for key in sorted(g):
        if key.startswith(name + "/o"):
            p = key.split("/")[1].split("_")
            # o<order> n<steps> <type: dpm solver[pp]> <method> <solver_type...> <skip...> d<denoise> [g<scale>]
            order, steps = int(p[0][1:]), int(p[1][1:])
            rest = "_".join(p[2:])
            typ = "dpm_solver++" if rest.startswith("dpm_solverpp") else "dpm_solver"
            rest = rest[len("dpm_solverpp_" if typ == "dpm_solver++" else "dpm_solver_"):]
            assert rest.startswith("singlestep_")
            rest = rest[len("singlestep_"):]
            solver_type = "dpm_solver" if rest.startswith("dpm_solver_") else "taylor"
            rest = rest[len(solver_type) + 1:]
            skip = "time_uniform" if rest.startswith("time_uniform") else "time_quadratic"
            gs = float(rest.split("_g")[1]) if "_g" in rest else None
            yield key, dict(order=order, steps=steps, type=typ, method="singlestep", solver_type=solver_type,
                            skip_type=skip, denoise=False), gs


def _rel(name, key):
    """Goldens whose reference values are not O(1) by construction: small_vp_score (golden_lib.TRAINED comment) and
    dpm_solver++ on the energy-guided models (the reference's guided data prediction is, as written, (-std x + cov
    score) / scale, energy_conditional_diffusion_model.py:1281-1306, which diverges)."""
    return name == "small_vp_score" or (name in ("small_energy", "qgpo_noise") and ("solverpp" in key or "solver++" in key))


def close(a, b, tol, what, rel=False):
    """Absolute max-abs tolerance (the DPM goldens of the trained VP models are O(1) actions); rel=True only for
    small_vp_score, whose reference trajectory diverges by construction (golden_lib.TRAINED comment)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    ref = float(np.abs(b).max())
    scale = max(1.0, ref) if rel else 1.0
    err = float(np.abs(a - b).max())
    line = "[parity] %-60s max-abs err %.3e  max|ref| %.3e  tol %.1e%s" % (what, err, ref, tol, " x max|ref|" if rel else " abs")
    print(line)
    assert err <= tol * scale, line


@pytest.mark.parametrize("name", MODELS)
def test_oracle_dpm_solver_vs_reference_golden(name):
    g = _load()
    kind, cfg = gl.model_zoo()[name]
    sd = gu.weights(name)
    path = O.Path(cfg["path"])
    mt = cfg["model"]["type"]
    x0, cond = gu.T(g[name + "/x0"]), gu.T(g[name + "/cond"])
    n = 0
    for key, kw, gs in _cases(g, name):
        net = lambda t, x: O.intrinsic_forward(sd, t, x, cond)  # noqa: E731
        if kind == "energy":
            score = lambda t, x: O.guided_score(sd, mt, path, t, x, cond, gs)  # noqa: E731
            noise_fn = lambda t, x: -score(t, x) * O._b(path.std(t), x)  # noqa: E731
            data_fn = lambda t, x: (-O._b(path.std(t), x) * x + O._b(path.cov(t), x) * score(t, x)) / O._b(path.scale(t), x)  # noqa: E731
        else:
            noise_fn = lambda t, x: O.noise_head(mt, path, net(t, x), t, x)  # noqa: E731
            data_fn = lambda t, x: O.data_head(mt, path, net(t, x), t, x)  # noqa: E731
        with torch.enable_grad():
            out = O.dpm_solver_integrate(path, noise_fn, data_fn, x0, **kw)
        close(out.detach(), g[key], 2e-5, name + " " + key, rel=_rel(name, key))
        n += 1
    assert n >= 6
    assert any(r.startswith(name + "/") for r in g["raises"])  # the reference's own failing option sets are recorded


def test_orders_match_reference_schedule():
    assert O.dpm_orders_singlestep(17, 2) == [2] * 8 + [1]          # QGPO: 8 second-order steps + 1 first-order = 17 NFE
    assert O.dpm_orders_singlestep(10, 2) == [2] * 4 + [1, 1]
    assert O.dpm_orders_singlestep(10, 3) == [3, 3, 3, 1]
    from generativerl_preview_amd.numerical_methods.numerical_solvers import DPMSolver, get_solver
    assert get_solver("DPMSolver") is DPMSolver
    for steps in range(3, 25):
        for order in (1, 2, 3):
            o = DPMSolver.get_orders_for_singlestep_solver(steps, order)
            assert o == O.dpm_orders_singlestep(steps, order) and sum(o) == steps


def test_unsupported_options_raise_like_the_reference():
    from generativerl_preview_amd.numerical_methods.numerical_solvers import DPMSolver
    from generativerl_preview_amd.generative_models.diffusion_process import DiffusionProcess
    from generativerl_preview_amd.numerical_methods.probability_path import GaussianConditionalProbabilityPath
    dp = DiffusionProcess(GaussianConditionalProbabilityPath(gl.VP))
    x = torch.zeros(2, 3)
    f = lambda t, x, c: x  # noqa: E731
    with pytest.raises(TypeError):
        DPMSolver(order=2, device="cpu", steps=6, skip_type="logSNR").integrate(dp, f, f, x)
    with pytest.raises(TypeError):
        DPMSolver(order=2, device="cpu", steps=6, denoise=True).integrate(dp, f, f, x)
    with pytest.raises(NotImplementedError):
        DPMSolver(order=2, device="cpu", steps=6, method="multistep").integrate(dp, f, f, x)
    # host logic with a trivial "network": matches the oracle bit for bit on the CPU
    nf = lambda t, x, c: 0.3 * x + t[:, None]  # noqa: E731
    got = DPMSolver(order=3, device="cpu", steps=10, solver_type="taylor").integrate(dp, nf, nf, torch.ones(2, 3), save_intermediate=True)
    want = O.dpm_solver_integrate(O.Path(gl.VP), lambda t, x: nf(t, x, None), lambda t, x: nf(t, x, None), torch.ones(2, 3),
                                  steps=10, order=3, solver_type="taylor")
    close(got, want, 1e-6, "host DPM-Solver vs oracle")


@pytest.mark.gpu
@pytest.mark.parametrize("name", MODELS)
def test_gpu_dpm_solver_vs_reference_golden(name):
    DEV = "cuda:0"
    g = _load()
    kind, cfg, m = um.build(name, DEV)
    x0 = torch.as_tensor(g[name + "/x0"]).to(DEV)
    cond = torch.as_tensor(g[name + "/cond"]).to(DEV)
    n = 0
    for key, kw, gs in _cases(g, name):
        sc = dict(type="DPMSolver", args=dict(device=DEV, **kw))
        call = dict(t_span=None, x_0=x0, condition=cond, solver_config=sc)
        if gs is not None:
            call["guidance_scale"] = gs
        traj = m.sample_forward_process(**call)
        close(traj.cpu(), g[key], 1e-4, name + " " + key, rel=_rel(name, key))
        close(m.sample(**call).cpu(), g[key][-1], 1e-4, name + " " + key + " final", rel=_rel(name, key))
        n += 1
    assert n >= 6


@pytest.mark.gpu
def test_gpu_fused_stages_equal_host_driven_stages_and_bf16_runs_in_the_persistent_kernel():
    """Every order and solver type goes through grlhip_sample_stages (one call; the persistent bf16 integrator for the
    QGPO shapes; the taylor form of order 3 uses the second saved model value):
    same result as the host-driven evaluation in fp32, within the bf16 tolerance of the oracle in bf16."""
    from generativerl_preview_amd import _lib as L
    from generativerl_preview_amd.numerical_methods.numerical_solvers import DPMSolver
    import generativerl_preview_amd.engine as engine
    DEV = "cuda:0"
    kind, cfg, m = um.build("qgpo_noise", DEV)
    sd = gu.weights("qgpo_noise")
    gen = torch.Generator().manual_seed(31)
    x0, cond = torch.randn(700, 6, generator=gen), torch.randn(700, 17, generator=gen)
    calls = []
    orig = engine.NetPlan.sample_stages
    engine.NetPlan.sample_stages = lambda self, *a, **k: (calls.append(k.get("dtype")), orig(self, *a, **k))[1]
    try:
        for typ, order, stype in (("dpm_solver++", 2, "dpm_solver"), ("dpm_solver", 2, "dpm_solver"), ("dpm_solver++", 3, "dpm_solver"),
                                  ("dpm_solver", 3, "dpm_solver"), ("dpm_solver", 3, "taylor"), ("dpm_solver++", 3, "taylor"),
                                  ("dpm_solver", 2, "taylor")):
            args = dict(order=order, device=DEV, steps=17, type=typ, solver_type=stype)
            n_before = len(calls)
            fused = m.sample_forward_process(x_0=x0.to(DEV), condition=cond.to(DEV), guidance_scale=1.0,
                                             solver_config=dict(type="DPMSolver", args=args))
            assert len(calls) == n_before + 1   # ONE grlhip_sample_stages call, the taylor form of order 3 included
            solver = DPMSolver(**args)
            solver.fused = False
            nf, df = m._dpm_functions(1.0)
            host = solver.integrate(m.diffusion_process, nf, df, x0.to(DEV), cond.to(DEV), save_intermediate=True)
            assert solver.nfe == 17
            close(fused.cpu(), host.cpu(), 2e-5, "fused vs host-driven %s order %d %s" % (typ, order, stype), rel="++" in typ)
        # bf16: the persistent integrator, on the O(1) noise-prediction form (the guided data-prediction form of
        # dpm_solver++ diverges as written in the reference, see _rel)
        path, mt = O.Path(cfg["path"]), cfg["model"]["type"]
        score = lambda t, x: O.guided_score(sd, mt, path, t, x, cond, 1.0)  # noqa: E731
        noise_fn = lambda t, x: -score(t, x) * O._b(path.std(t), x)  # noqa: E731
        data_fn = lambda t, x: (-O._b(path.std(t), x) * x + O._b(path.cov(t), x) * score(t, x)) / O._b(path.scale(t), x)  # noqa: E731
        with torch.enable_grad():
            want = O.dpm_solver_integrate(path, noise_fn, data_fn, x0, steps=17, order=2, type="dpm_solver")
        m.model.compute_dtype = L.BF16
        n0 = len(calls)
        got = m.sample_forward_process(x_0=x0.to(DEV), condition=cond.to(DEV), guidance_scale=1.0,
                                       solver_config=dict(type="DPMSolver", args=dict(order=2, device=DEV, steps=17, type="dpm_solver")))
        assert calls[n0:] == [L.BF16]
        close(got.cpu(), want.detach(), 2e-2, "bf16 fused DPM-Solver-2 trajectory", rel=True)
        close(m.sample(x_0=x0.to(DEV), condition=cond.to(DEV), guidance_scale=1.0,
                       solver_config=dict(type="DPMSolver", args=dict(order=2, device=DEV, steps=17, type="dpm_solver"))).cpu(),
              want[-1].detach(), 2e-2, "bf16 fused DPM-Solver-2 final")
    finally:
        engine.NetPlan.sample_stages = orig


@pytest.mark.parametrize("name,typ,order,steps,stype", [
    ("small_vp_noise", "dpm_solver", 2, 17, "dpm_solver"), ("small_vp_velocity", "dpm_solver++", 2, 10, "dpm_solver"),
    ("small_energy", "dpm_solver", 3, 10, "dpm_solver"), ("small_energy", "dpm_solver++", 3, 9, "dpm_solver"),
    ("small_vp_score", "dpm_solver++", 1, 6, "dpm_solver"), ("small_vp_noise", "dpm_solver", 3, 10, "taylor"),
    ("small_energy", "dpm_solver++", 3, 9, "taylor"), ("small_vp_velocity", "dpm_solver", 2, 8, "taylor")])
def test_stage_table_reproduces_the_solver_on_the_cpu(name, typ, order, steps, stype):
    """Host logic of the fused path, no GPU: the stage table DPMSolver hands to grlhip_sample_stages, executed by a
    ten-line interpreter over the ORACLE's network (and energy gradient), gives the oracle's DPM-Solver trajectory."""
    from generativerl_preview_amd.numerical_methods.numerical_solvers import DPMSolver
    kind, cfg, m = um.build(name)
    sd = gu.weights(name)
    gen = torch.Generator().manual_seed(2)
    x0, cond = torch.randn(9, 6, generator=gen), torch.randn(9, 17, generator=gen)
    solver = DPMSolver(order=order, device="cpu", steps=steps, type=typ, solver_type=stype)
    gs = 1.0 if kind == "energy" else None
    noise_fn, data_fn = m._dpm_functions(gs)
    fn = data_fn if typ == "dpm_solver++" else noise_fn
    orders = solver.get_orders_for_singlestep_solver(steps=steps, order=order)
    ts = solver.get_time_steps(t_T=float(m.diffusion_process.t_max), t_0=0.00001, N=steps)
    table = solver.stage_table(m.diffusion_process, fn, orders, ts)
    assert table.shape == (steps, 12) and int(table[:, 9].max()) == len(orders)
    x, xs, ms, ms2, traj = x0, torch.zeros_like(x0), torch.zeros_like(x0), torch.zeros_like(x0), [x0]
    for row in table.tolist():
        tau, P, Q, R, W, cn, cx, cg, save, emit, W2 = row[:11]
        t = torch.full((x.shape[0],), tau)
        mval = cn * O.intrinsic_forward(sd, t, x, cond) + cx * x
        if kind == "energy":
            mval = mval + cg * O.energy_guidance(sd, t, x, cond, 1.0)
        xn = P * x + Q * xs + R * mval + W * ms + W2 * ms2
        if save == 1:
            xs, ms = x, mval
        elif save == 2:
            ms2 = mval
        x = xn
        if emit >= 0:
            traj.append(x)
    path, mt = O.Path(cfg["path"]), cfg["model"]["type"]
    if kind == "energy":
        score = lambda t, y: O.guided_score(sd, mt, path, t, y, cond, 1.0)  # noqa: E731
        nf = lambda t, y: -score(t, y) * O._b(path.std(t), y)  # noqa: E731
        df = lambda t, y: (-O._b(path.std(t), y) * y + O._b(path.cov(t), y) * score(t, y)) / O._b(path.scale(t), y)  # noqa: E731
    else:
        net = lambda t, y: O.intrinsic_forward(sd, t, y, cond)  # noqa: E731
        nf = lambda t, y: O.noise_head(mt, path, net(t, y), t, y)  # noqa: E731
        df = lambda t, y: O.data_head(mt, path, net(t, y), t, y)  # noqa: E731
    want = O.dpm_solver_integrate(path, nf, df, x0, steps=steps, order=order, type=typ, solver_type=stype)
    got = torch.stack(traj)
    assert got.shape == want.shape
    close(got.detach(), want.detach(), 2e-4, "stage table vs oracle solver", rel=_rel(name, typ))
