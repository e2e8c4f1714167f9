"""GPU parity: the HIP path (through the C ABI) against the committed goldens and against the CPU
oracle on larger seeded inputs.

Tolerances (north_star): final actions / trajectories ABSOLUTE max-abs 1e-4 (fp32 engine) and 2e-2 (bf16 engine) -- the
VP golden models carry trained weights (tests/golden/make_trained_weights.py) so that sampled actions are O(1) and
an absolute criterion means something; losses 1e-5 relative.  ``rel=True`` (tolerance x max(1, max|ref|)) is used
only where the checked quantity is not an action: gradients (scale of the tensor), head conversions / guided scores
(they carry 1/sigma factors) and the three goldens whose reference trajectories diverge by construction
(golden_lib.TRAINED comment: small_vp_score, small_vp_data, small_gvp_score).  Every comparison prints its achieved
error and appends it to gpurun_out/parity_errors.log."""
import os

import numpy as np
import pytest
import torch

import golden_lib as gl
import golden_util as gu
import util_models as um
from generativerl_preview_amd import _lib as L
from oracle import grl_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ALL = gu.CANON + gu.SMALL
TOL = 1e-4  # north_star: fp32 final actions within max-abs 1e-4


REL_TRAJ = ("small_vp_score", "small_vp_data", "small_gvp_score")  # reference trajectories diverge by construction
_LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_errors.log")


def close(a, b, tol=TOL, what="", rel=False):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    ref = float(np.max(np.abs(b))) if b.size else 0.0
    scale = max(1.0, ref) if rel else 1.0
    err = float(np.max(np.abs(a - b))) if b.size else 0.0
    line = "[parity] %-60s max-abs err %.3e  max|ref| %.3e  tol %.1e%s  %s" % (
        what, err, ref, tol, " x max|ref|" if rel else " abs", os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0])
    print(line)
    try:
        os.makedirs(os.path.dirname(_LOG), exist_ok=True)
        with open(_LOG, "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
    assert err <= tol * scale, line


def cu(a):
    return None if a is None else torch.as_tensor(np.asarray(a)).to(DEV)


def test_library_loaded_and_device_visible():
    assert L.device_available()


@pytest.mark.parametrize("name", ALL)
def test_net_forward_vs_golden(name):
    kind, cfg, m = um.build(name, DEV)
    g = gu.load("model_%s.npz" % name)
    out = m.model(cu(g["t"]), cu(g["x"]), cu(g.get("cond")))
    close(out, g["net"], what=name + " net")


@pytest.mark.parametrize("name", [n for n in ALL if not n.endswith("icfm") and n != "icfm_cond"])
def test_head_conversions_vs_golden(name):
    """DiffusionModel.score_function / noise_function / velocity_function / data_prediction_function (HIP forward +
    the schedule algebra as written in the reference); a head the reference itself cannot produce must raise."""
    kind, cfg, m = um.build(name, DEV)
    g = gu.load("model_%s.npz" % name)
    t, x, c = cu(g["t"]), cu(g["x"]), cu(g.get("cond"))
    n = 0
    for h, fn in (("score", m.score_function), ("noise", m.noise_function), ("velocity", m.velocity_function),
                  ("data", m.data_prediction_function)):
        if "head_" + h in g:
            close(fn(t, x, c), g["head_" + h], tol=2e-4, what=name + " head_" + h, rel=True)
            n += 1
        else:
            with pytest.raises(NotImplementedError):
                fn(t, x, c)
    assert n >= 3


@pytest.mark.parametrize("name", ["qgpo_noise", "small_energy", "hopper_noise"])
def test_energy_guidance_vs_golden(name):
    kind, cfg, m = um.build(name, DEV)
    g = gu.load("model_%s.npz" % name)
    t, x, c = cu(g["t"]), cu(g["x"]), cu(g["cond"])
    close(m.energy_guidance(t, x, c), g["energy"], what=name + " energy")
    for s in (0.0, 1.0, 5.0):
        close(m.energy_guidance.calculate_energy_guidance(t, x, c, s), g["guidance_%g" % s], what=name + " guidance")
        close(m.score_function_with_energy_guidance(t, x, c, s), g["guided_score_%g" % s], what=name + " guided score", rel=True)


def _cases(s):
    for key in sorted(s):
        if key.startswith("traj_"):
            parts = key.split("_")
            yield (key, int(parts[1][1:]), None if parts[2] == "bNone" else int(parts[2][1:]), parts[3] == "x1",
                   float(parts[4][1:]) if len(parts) > 4 else None)


@pytest.mark.parametrize("name", ALL)
def test_ode_trajectories_vs_golden(name):
    """All four (batch_size, x_0) combinations, T in {5, 32, 101}: shapes, layout quirk and values."""
    kind, cfg, m = um.build(name, DEV)
    s = gu.load("sample_%s.npz" % name)
    S = gu.cond_dim(cfg)
    n = 0
    for key, Tn, bs, use_x0, gs in _cases(s):
        kw = dict(t_span=torch.linspace(0.0, 1.0, Tn), batch_size=bs, x_0=cu(s["x0"]) if use_x0 else None)
        if S:
            kw["condition"] = cu(s["cond"])
        if gs is not None:
            kw["guidance_scale"] = gs
        if not use_x0:
            um.fix_noise(m, cu(s["xT_T%d_b%s_x0" % (Tn, bs)]))
        traj = m.sample_forward_process(**kw)
        close(traj, s[key], what=name + " " + key, rel=name in REL_TRAJ)
        final = m.sample(**kw)
        close(final, s[key][-1], what=name + " " + key + " final", rel=name in REL_TRAJ)
        n += 1
    assert n >= 4


@pytest.mark.parametrize("name", ["small_energy", "small_uncond_noise", "swiss_noise", "qgpo_noise"])
def test_sde_trajectories_vs_golden(name):
    kind, cfg, m = um.build(name, DEV)
    s = gu.load("sample_%s.npz" % name)
    S = gu.cond_dim(cfg)
    n = 0
    for Tn, dt in ((11, 0.1), (21, 0.05)):
        for key in [k for k in s if k.startswith("sde_T%d" % Tn) and not k.endswith(("dW", "xT"))]:
            kw = dict(t_span=torch.linspace(0.0, 1.0, Tn), batch_size=2,
                      solver_config=dict(type="SDESolver", args=dict(library="torchsde", dt=dt)),
                      noise_increments=cu(s["sde_T%d_dW" % Tn]))
            if S:
                kw["condition"] = cu(s["cond"])
            if "_g" in key:
                kw["guidance_scale"] = float(key.split("_g")[1])
            um.fix_noise(m, cu(s["sde_T%d_xT" % Tn]))
            close(m.sample_forward_process(**kw), s[key], what=name + " " + key)
            n += 1
    assert n >= 2


@pytest.mark.parametrize("name,rows", [("qgpo_noise", 1000), ("icfm_cond", 777), ("gvp_wide", 300),
                                       ("swiss_noise", 500), ("small_vp_data", 97), ("small_gvp_score", 65)])
def test_ode_sampling_vs_oracle_ragged_rows(name, rows):
    """Row counts that are not multiples of the 32-row tile; 20 Euler steps; final actions."""
    kind, cfg, m = um.build(name, DEV)
    sd = gu.weights(name)
    S = gu.cond_dim(cfg)
    D = cfg["x_size"]
    gen = torch.Generator().manual_seed(5)
    x0 = torch.randn(rows, D, generator=gen)
    cond = torch.randn(rows, S, generator=gen) if S else None
    lo, hi = (1e-3, 1.0 - 1e-3) if name == "small_gvp_score" else (0.0, 1.0)
    t_span = torch.linspace(lo, hi, 21)
    kw = dict(t_span=t_span, x_0=x0.to(DEV))
    if S:
        kw["condition"] = cond.to(DEV)
    if kind == "energy":
        kw["guidance_scale"] = 1.5
    got = m.sample(**kw)
    want = O.sample_forward_process(sd, cfg, kind, t_span, x_0=x0, condition=cond, guidance_scale=1.5)[-1]
    assert got.shape == want.shape
    close(got, want, what=name + " 20-step finals vs oracle", rel=name in REL_TRAJ)


def test_empty_batch_and_errors():
    kind, cfg, m = um.build("small_vp_noise", DEV)
    out = m.sample(t_span=torch.linspace(0, 1, 5), x_0=torch.zeros(0, 6, device=DEV),
                   condition=torch.zeros(0, 17, device=DEV))
    assert out.shape == (0, 6)
    with pytest.raises(AssertionError):
        m.sample(t_span=torch.linspace(0, 1, 5), x_0=torch.zeros(3, 5, device=DEV), condition=torch.zeros(3, 17, device=DEV))
    with pytest.raises(L.GrlHipError):  # conditional net without a condition
        m.sample(t_span=torch.linspace(0, 1, 5), x_0=torch.zeros(3, 6, device=DEV))


def test_weights_are_repacked_after_an_update():
    kind, cfg, m = um.build("small_uncond_noise", DEV)
    g = gu.load("model_small_uncond_noise.npz")
    t, x = cu(g["t"]), cu(g["x"])
    a = m.model(t, x)
    with torch.no_grad():
        m.model.model["backbone"].last_block.bias.add_(1.0)
    b = m.model(t, x)
    close(b - a, torch.ones_like(a), what="bias shift")


# ----------------------------------------------------------------------------- bf16 MFMA engine
BF16_TOL = 2e-2  # north_star: bf16 final actions within max-abs 2e-2


def _bf16_model(name):
    import generativerl_preview_amd as G
    kind, cfg = gl.model_zoo()[name]
    cfg = dict(cfg, device=DEV)
    cfg.pop("energy_guidance", None)
    m = (G.IndependentConditionalFlowModel if kind == "icfm" else G.DiffusionModel)(cfg)
    sd = {k: v for k, v in gu.weights(name).items() if k.startswith("model.")}
    m.load_state_dict(sd)
    m = m.to(DEV)
    m.model.compute_dtype = L.BF16
    return ("icfm" if kind == "icfm" else "diffusion"), cfg, m, sd


@pytest.mark.parametrize("name,rows,steps", [("qgpo_noise", 1000, 100), ("icfm_cond", 300, 100),
                                             ("swiss_noise", 500, 100), ("qgpo_noise", 128 * 300 + 5, 10),
                                             ("hopper_noise", 1000, 100), ("wideobs_noise", 700, 100)])
def test_bf16_ode_sampling_vs_oracle(name, rows, steps):
    """Every [512,256,128] net runs the persistent fused integrator, whatever its action / observation widths (shape
    CLASSES with run-time widths: Hopper 3/11, a 44-wide observation = three condition k-steps)."""
    kind, cfg, m, sd = _bf16_model(name)
    assert "k_sample_bf16" in m.model.plan.fused_kernel(), "bf16 sampling of %s would not run the fused integrator" % name
    S = gu.cond_dim(cfg)
    D = cfg["x_size"]
    gen = torch.Generator().manual_seed(11)
    x0 = torch.randn(rows, D, generator=gen)
    cond = torch.randn(rows, S, generator=gen) if S else None
    t_span = torch.linspace(0.0, 1.0, steps + 1)
    kw = dict(t_span=t_span, x_0=x0.to(DEV))
    if S:
        kw["condition"] = cond.to(DEV)
    got = m.sample(**kw).cpu()
    chunk = slice(0, min(rows, 2048))  # the oracle is slow: check a prefix and the ragged tail
    tail = slice(max(0, rows - 64), rows)
    for sl in (chunk, tail):
        want = O.sample_forward_process(sd, cfg, kind, t_span, x_0=x0[sl], condition=None if cond is None else cond[sl])[-1]
        close(got[sl], want, tol=BF16_TOL, what="bf16 %s %d rows %d steps finals" % (name, rows, steps))


def test_bf16_trajectory_and_sde_vs_golden():
    """swiss-roll net (unconditional): golden ODE trajectories and Euler-Maruyama with fixed increments."""
    kind, cfg, m, sd = _bf16_model("swiss_noise")
    s = gu.load("sample_swiss_noise.npz")
    for key, Tn, bs, use_x0, gs in _cases(s):
        kw = dict(t_span=torch.linspace(0.0, 1.0, Tn), batch_size=bs, x_0=cu(s["x0"]) if use_x0 else None)
        if not use_x0:
            um.fix_noise(m, cu(s["xT_T%d_b%s_x0" % (Tn, bs)]))
        traj = m.sample_forward_process(**kw)
        close(traj, s[key], tol=BF16_TOL, what="bf16 swiss trajectory " + key)
        close(traj[-1], s[key][-1], tol=BF16_TOL, what="bf16 swiss final " + key)
    for Tn, dt in ((11, 0.1), (21, 0.05)):
        um.fix_noise(m, cu(s["sde_T%d_xT" % Tn]))
        out = m.sample_forward_process(t_span=torch.linspace(0.0, 1.0, Tn), batch_size=2,
                                       solver_config=dict(type="SDESolver", args=dict(library="torchsde", dt=dt)),
                                       noise_increments=cu(s["sde_T%d_dW" % Tn]))
        close(out, s["sde_T%d" % Tn], tol=BF16_TOL, what="bf16 swiss sde trajectory T%d" % Tn)
        close(out[-1], s["sde_T%d" % Tn][-1], tol=BF16_TOL, what="bf16 swiss sde final T%d" % Tn)


def _bf16_energy_model(name="qgpo_noise"):
    kind, cfg, m = um.build(name, DEV)
    m.model.compute_dtype = L.BF16
    return cfg, m, gu.weights(name)


def test_bf16_guided_sampling_vs_golden():
    """QGPO net + energy net fused in the bf16 integrator: guided ODE trajectories (every batch_size / x_0
    combination, guidance_scale 0 and 1) and guided Euler-Maruyama with the golden increments."""
    cfg, m, sd = _bf16_energy_model()
    s = gu.load("sample_qgpo_noise.npz")
    n = 0
    for key, Tn, bs, use_x0, gs in _cases(s):
        kw = dict(t_span=torch.linspace(0.0, 1.0, Tn), batch_size=bs, x_0=cu(s["x0"]) if use_x0 else None,
                  condition=cu(s["cond"]), guidance_scale=gs)
        if not use_x0:
            um.fix_noise(m, cu(s["xT_T%d_b%s_x0" % (Tn, bs)]))
        close(m.sample_forward_process(**kw), s[key], tol=BF16_TOL, what="bf16 guided trajectory " + key)
        close(m.sample(**kw), s[key][-1], tol=BF16_TOL, what="bf16 guided final " + key)
        n += 1
    assert n >= 8
    for Tn, dt in ((11, 0.1), (21, 0.05)):
        for gs in (0, 1):
            um.fix_noise(m, cu(s["sde_T%d_xT" % Tn]))
            out = m.sample_forward_process(t_span=torch.linspace(0.0, 1.0, Tn), batch_size=2, condition=cu(s["cond"]),
                                           guidance_scale=float(gs),
                                           solver_config=dict(type="SDESolver", args=dict(library="torchsde", dt=dt)),
                                           noise_increments=cu(s["sde_T%d_dW" % Tn]))
            close(out, s["sde_T%d_g%d" % (Tn, gs)], tol=BF16_TOL, what="bf16 guided sde trajectory T%d g%d" % (Tn, gs))
            close(out[-1], s["sde_T%d_g%d" % (Tn, gs)][-1], tol=BF16_TOL, what="bf16 guided sde final T%d g%d" % (Tn, gs))


@pytest.mark.parametrize("name,rows,steps,scale", [("qgpo_noise", 1000, 100, 10.0), ("qgpo_noise", 128 * 260 + 37, 10, 30.0),
                                                   ("hopper_noise", 900, 100, 10.0)])
def test_bf16_guided_ode_vs_oracle(name, rows, steps, scale):
    cfg, m, sd = _bf16_energy_model(name)
    D, S = cfg["x_size"], gu.cond_dim(cfg)
    gen = torch.Generator().manual_seed(23)
    x0 = torch.randn(rows, D, generator=gen)
    cond = torch.randn(rows, S, generator=gen)
    t_span = torch.linspace(0.0, 1.0, steps + 1)
    got = m.sample(t_span=t_span, x_0=x0.to(DEV), condition=cond.to(DEV), guidance_scale=scale).cpu()
    fk = m.model.plan.fused_kernel(m.energy_guidance.model.plan)
    assert "k_sample_bf16" in fk and "energy-guided" in fk, fk
    worst = 0.0
    for sl in (slice(0, min(rows, 1024)), slice(max(0, rows - 64), rows)):
        want = O.sample_forward_process(sd, cfg, "energy", t_span, x_0=x0[sl], condition=cond[sl], guidance_scale=scale)[-1]
        close(got[sl], want, tol=BF16_TOL, what="bf16 guided %s %d rows %d steps scale %g" % (name, rows, steps, scale))
        worst = max(worst, float((got[sl] - want).abs().max()))
    # the guidance term must matter at this scale (otherwise the test proves nothing): leaving it out moves the
    # actions by several times the error the guided kernel makes
    un = O.sample_forward_process(sd, cfg, "energy", t_span, x_0=x0[:64], condition=cond[:64], guidance_scale=0.0)[-1]
    gd = O.sample_forward_process(sd, cfg, "energy", t_span, x_0=x0[:64], condition=cond[:64], guidance_scale=scale)[-1]
    effect = float(np.max(np.abs(np.asarray(un) - np.asarray(gd))))
    print("[parity] guidance effect %.3e vs worst bf16 error %.3e" % (effect, worst))
    assert effect > 4 * worst and effect > BF16_TOL, (effect, worst)


@pytest.mark.parametrize("name,rows,steps", [("small_vp_noise", 333, 20), ("gvp_wide", 300, 20), ("small_uncond_noise", 130, 50),
                                             ("small_icfm", 257, 20)])
def test_bf16_generic_shapes_run_on_the_batch_engine(name, rows, steps):
    """Shapes without a fused bf16 kernel run layer by layer: C5's [1024] net on the row-major bf16 engine of the wide
    nets (k_w_gemm: bf16 activations in HBM), the small nets on the batch engine with bf16 MFMAs (fp32 activations in
    HBM, bf16 operands, fp32 accumulate) -- same 2e-2 tolerance."""
    kind, cfg, m, sd = _bf16_model(name)
    assert m.model.plan.fused_kernel() == ""
    assert ("k_w_gemm" if name == "gvp_wide" else "k_bt_gemm_bf16") in m.model.plan.bf16_engine()
    S = gu.cond_dim(cfg)
    D = cfg["x_size"]
    gen = torch.Generator().manual_seed(17)
    x0 = torch.randn(rows, D, generator=gen)
    cond = torch.randn(rows, S, generator=gen) if S else None
    t_span = torch.linspace(0.0, 1.0, steps + 1)
    kw = dict(t_span=t_span, x_0=x0.to(DEV))
    if S:
        kw["condition"] = cond.to(DEV)
    got = m.sample(**kw).cpu()
    want = O.sample_forward_process(sd, cfg, kind, t_span, x_0=x0, condition=cond)[-1]
    close(got, want, tol=BF16_TOL, what="bf16 layer engine %s" % name)
    m.model.compute_dtype = L.F32
    exact = m.sample(**kw).cpu()
    assert float((got - exact).abs().max()) > 0.0, "bf16 path returned the fp32 result bit for bit: not the bf16 kernels"


def test_bf16_wide_layers_256_tiles_match_the_128_tiles_bit_for_bit():
    """C5's [1024] net on the wide engine (k_w_gemm, persistent 256 x 256 tiles) at 65 536, 1 024 and 65 836 rows: rows
    are independent and the k order of a row does not depend on the batch, so the first 1 024 rows of the big batch must
    equal the small batch BIT FOR BIT (the small batch is the one the oracle test above covers), and a ragged row count
    (padded to whole 256-row tiles) must agree with the big batch on the rows they share.  (Written in round 2 for the
    batch engine's 256-tile against its 128-tile kernel; that pair is exercised by the three-level net below.)"""
    kind, cfg, m, sd = _bf16_model("gvp_wide")
    assert "k_w_gemm" in m.model.plan.bf16_engine()
    gen = torch.Generator().manual_seed(23)
    rows = 65536 + 300
    x0 = torch.randn(rows, cfg["x_size"], generator=gen).to(DEV)
    t_span = torch.linspace(0.0, 1.0, 6)
    small = m.sample(t_span=t_span, x_0=x0[:1024].contiguous())
    big = m.sample(t_span=t_span, x_0=x0[:65536].contiguous())
    ragged = m.sample(t_span=t_span, x_0=x0)
    assert torch.isfinite(big).all() and float(big.abs().max()) > 0
    assert torch.equal(big[:1024], small)
    assert torch.equal(ragged[:65536], big)
    want = O.sample_forward_process(sd, cfg, kind, t_span, x_0=x0[65536 - 64:65536 + 64].cpu())[-1]
    close(ragged[65536 - 64:65536 + 64].cpu(), want, tol=BF16_TOL, what="bf16 wide engine, rows around 65536")


def test_bf16_256_tiles_on_a_three_level_net_without_a_fused_kernel():
    """A [512,256,128] net with 20 action dimensions (no fused integrator: the batch engine in bf16) at 140 000 rows: its
    512- and 256-wide layers (K = 128 ... 1024, dual products, U-Net concatenations) run the 256-tile kernel, the 128-wide
    ones and the whole net at 1 000 rows the 128-tile kernel.  Rows are independent: bit-identical results."""
    import generativerl_preview_amd as G
    cfg = dict(gl.diffusion_cfg(20, gl.VP, "noise_function", [512, 256, 128], 17), device=DEV)
    torch.manual_seed(20)
    m = G.DiffusionModel(cfg).to(DEV)
    m.model.compute_dtype = L.BF16
    assert m.model.plan.fused_kernel() == "" and "k_bt_gemm_bf16" in m.model.plan.bf16_engine()
    gen = torch.Generator().manual_seed(24)
    rows = 140000
    x0 = torch.randn(rows, 20, generator=gen).to(DEV)
    cond = torch.randn(rows, 17, generator=gen).to(DEV)
    t_span = torch.linspace(0.0, 1.0, 4)
    big = m.sample(t_span=t_span, x_0=x0, condition=cond)
    small = m.sample(t_span=t_span, x_0=x0[:1000].contiguous(), condition=cond[:1000].contiguous())
    tail = m.sample(t_span=t_span, x_0=x0[-777:].contiguous(), condition=cond[-777:].contiguous())
    assert torch.isfinite(big).all() and float(big.abs().max()) > 0
    assert torch.equal(big[:1000], small) and torch.equal(big[-777:], tail)


@pytest.mark.parametrize("x_dim,c_dim,guided", [(1, 5, False), (16, 32, True), (16, 33, False), (8, 16, True), (5, 67, False),
                                                 (16, None, False), (1, None, False)])
def test_bf16_fused_kernel_at_the_edges_of_its_shape_classes(x_dim, c_dim, guided):
    """Action widths 1 and 16, observation widths around the 16-column k-steps (16, 32, 33, 67), unconditional: the padded
    shape classes of the fused integrator against the oracle.  Seeded (untrained) weights and 8 steps, so states grow:
    the first comparison is relative to max|ref| (2e-2) -- it checks the padding / run-time width plumbing, not the bf16
    budget; the guided classes are then run again from a start 200 x smaller, which ends at |x| = O(1), against the
    ABSOLUTE gates."""
    import generativerl_preview_amd as G
    cfg = dict(gl.diffusion_cfg(x_dim, gl.VP, "noise_function", [512, 256, 128], c_dim, energy=guided), device=DEV)
    torch.manual_seed(100 * x_dim + (c_dim or 0))
    m = (G.EnergyConditionalDiffusionModel if guided else G.DiffusionModel)(cfg).to(DEV)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if "last_block" in k:
                p.mul_(0.1)
    m.model.compute_dtype = L.BF16
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    rows = 300
    gen = torch.Generator().manual_seed(3)
    x0 = torch.randn(rows, x_dim, generator=gen)
    cond = torch.randn(rows, c_dim, generator=gen) if c_dim else None
    t_span = torch.linspace(0.0, 1.0, 9)
    kw = dict(t_span=t_span, x_0=x0.to(DEV))
    if c_dim:
        kw["condition"] = cond.to(DEV)
    if guided:
        kw["guidance_scale"] = 2.0
    got = m.sample(**kw).cpu()
    fk = m.model.plan.fused_kernel(m.energy_guidance.model.plan if guided else None)
    assert "k_sample_bf16" in fk and (("energy-guided" in fk) == guided), fk
    okw = dict(guidance_scale=2.0) if guided else {}
    want = O.sample_forward_process(sd, cfg, "energy" if guided else "diffusion", t_span, x_0=x0, condition=cond, **okw)[-1]
    close(got, want, tol=BF16_TOL, what="bf16 class edge x%d c%s guided %d" % (x_dim, c_dim, guided), rel=True)
    m.model.compute_dtype = L.F32
    exact = m.sample(**kw).cpu()
    close(exact, want, tol=TOL, what="fp32 class edge x%d c%s guided %d" % (x_dim, c_dim, guided), rel=True)
    # the same chain started 200 x smaller ends at |x| = O(1): ABSOLUTE gates for the guided classes too (the unguided
    # ones have theirs on an ICFM policy, test_bf16_shape_class_edges_absolute_icfm)
    if guided:
        kw["x_0"] = (x0 / 200.0).to(DEV)
        want_s = O.sample_forward_process(sd, cfg, "energy", t_span, x_0=x0 / 200.0, condition=cond, **okw)[-1]
        assert 0.05 < float(want_s.abs().max()) < 20.0, float(want_s.abs().max())
        close(m.sample(**kw).cpu(), want_s, tol=TOL, what="fp32 class edge (small start, absolute) x%d c%s" % (x_dim, c_dim))
        m.model.compute_dtype = L.BF16
        close(m.sample(**kw).cpu(), want_s, tol=BF16_TOL, what="bf16 class edge (small start, absolute) x%d c%s" % (x_dim, c_dim))


@pytest.mark.parametrize("x_dim,c_dim", [(1, 5), (16, 32), (16, 33), (8, 16), (5, 67), (16, None), (1, None), (9, 17)])
def test_bf16_shape_class_edges_absolute_icfm(x_dim, c_dim):
    """The same padded shape classes with an ABSOLUTE 2e-2 gate on O(1) actions: an ICFM policy (velocity net, Euler from
    x_0 ~ N(0, 1)) keeps |x| = O(1) with seeded weights, and it runs the same fused kernel body -- only the per-step
    coefficient table differs from the VP case above.  Covers x_dim 9 and 16 (> 8: the state's single-rounding path) and
    1, 5, 8 (hi + lo split)."""
    import generativerl_preview_amd as G
    cfg = dict(gl.icfm_cfg(x_dim, [512, 256, 128], c_dim), device=DEV)
    torch.manual_seed(7 + 100 * x_dim + (c_dim or 0))
    m = G.IndependentConditionalFlowModel(cfg).to(DEV)
    m.model.compute_dtype = L.BF16
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    rows = 300
    gen = torch.Generator().manual_seed(5)
    x0 = torch.randn(rows, x_dim, generator=gen)
    cond = torch.randn(rows, c_dim, generator=gen) if c_dim else None
    t_span = torch.linspace(0.0, 1.0, 21)
    kw = dict(t_span=t_span, x_0=x0.to(DEV))
    if c_dim:
        kw["condition"] = cond.to(DEV)
    got = m.sample(**kw).cpu()
    assert "k_sample_bf16" in m.model.plan.fused_kernel(None)
    want = O.sample_forward_process(sd, cfg, "icfm", t_span, x_0=x0, condition=cond)[-1]
    assert 0.3 < float(want.abs().max()) < 20.0          # O(1) actions that did move
    assert float((want - x0).abs().max()) > 0.05
    close(got, want, tol=BF16_TOL, what="bf16 class edge (icfm, absolute) x%d c%s" % (x_dim, c_dim))
    m.model.compute_dtype = L.F32
    close(m.sample(**kw).cpu(), want, tol=TOL, what="fp32 class edge (icfm, absolute) x%d c%s" % (x_dim, c_dim))


def test_bf16_full_size_properties():
    """BASELINE config C2 at its full per-GPU size (1 048 576 states x 32 candidates, 100 steps, + a ragged tail), where
    the oracle is out of reach: size-independent properties.  (1) determinism: two launches agree bit for bit; (2) row
    independence: any subset of rows run alone in a small launch gives bit-identical actions (a row's result must not
    depend on its position, its tile mates or the workgroup that got it); (3) a handful of those rows against the oracle
    at the absolute bf16 tolerance."""
    kind, cfg, m, sd = _bf16_model("qgpo_noise")
    rows = 1048576 * 32 + 37
    gen = torch.Generator(device=DEV).manual_seed(77)
    x0 = torch.randn(rows, 6, device=DEV, generator=gen)
    cond = torch.randn(rows, 17, device=DEV, generator=gen)
    t_span = torch.linspace(0.0, 1.0, 101)
    a = m.sample(t_span=t_span, x_0=x0, condition=cond)
    assert a.shape == (rows, 6) and bool(torch.isfinite(a).all())
    b = m.sample(t_span=t_span, x_0=x0, condition=cond)
    assert torch.equal(a, b), "two launches over the same inputs differ"
    del b
    pick = torch.cat([torch.randint(0, rows, (4000,), generator=torch.Generator().manual_seed(5)),
                      torch.arange(rows - 70, rows), torch.arange(0, 26)]).to(DEV)
    sub = m.sample(t_span=t_span, x_0=x0[pick].contiguous(), condition=cond[pick].contiguous())
    assert torch.equal(sub, a[pick]), "rows computed alone differ from the same rows inside the full launch"
    few = pick[-120:].cpu()
    want = O.sample_forward_process(sd, cfg, kind, t_span, x_0=x0[few.to(DEV)].cpu(), condition=cond[few.to(DEV)].cpu())[-1]
    close(a[few.to(DEV)].cpu(), want, tol=BF16_TOL, what="bf16 full-size launch, 120 rows vs oracle")


def test_bf16_mixed_pair_is_rejected():
    """small net (batch engine, fp32-layout image) + the [256,256] energy net (fused-kernel fragment image): no
    bf16 kernel takes that pair; the library must say so instead of reading an image in the wrong layout."""
    kind, cfg, m = um.build("small_energy", DEV)
    m.model.compute_dtype = L.BF16
    with pytest.raises(L.GrlHipError, match="bf16"):
        m.sample(t_span=torch.linspace(0, 1, 5), x_0=torch.zeros(4, 6, device=DEV), condition=torch.zeros(4, 17, device=DEV),
                 guidance_scale=1.0)


# ----------------------------------------------------------------------------- fused training step
def _check_grads(m, g, pfx):
    n = 0
    for k, p in m.named_parameters():
        if not k.startswith("model.") or k.endswith("t_encoder.W"):
            continue
        assert p.grad is not None, k
        if pfx + "_grad/" + k in g:
            close(p.grad, g[pfx + "_grad/" + k], tol=2e-4, what=pfx + " grad " + k, rel=True)
            n += 1
        elif pfx + "_gradsub/" + k in g:
            flat = p.grad.reshape(-1)
            close(flat[:: max(1, flat.numel() // 509)][:509], g[pfx + "_gradsub/" + k], tol=2e-4, what=pfx + " grad " + k, rel=True)
            s = g[pfx + "_gradsum/" + k]
            assert abs(float(flat.double().pow(2).sum().sqrt()) - s[1]) <= 2e-4 * s[1] + 1e-7, k
            n += 1
    assert n > 10


@pytest.mark.parametrize("name", ALL)
def test_losses_and_grads_vs_golden(name):
    """score_matching_loss / flow_matching_loss forward + backward: loss within 1e-5 relative
    (north_star), per-row losses and every parameter gradient against the reference's autograd."""
    kind, cfg, m = um.build(name, DEV)
    g = gu.load("model_%s.npz" % name)
    x, cond = cu(g["x"]), cu(g.get("cond"))
    cases = []
    if kind == "icfm":
        cases.append(("fm", lambda avg: m.flow_matching_loss(cu(g["eps"]), x, cond, average=avg, t=cu(g["fm_t"]), noise=cu(g["fm_eps"]))))
    else:
        if "sm_loss" in g:
            cases.append(("sm", lambda avg: m.score_matching_loss(x, cond, weighting_scheme="vanilla", average=avg,
                                                                  t=cu(g["sm_t"]), noise=cu(g["sm_eps"]))))
        if "fm_loss" in g:
            cases.append(("fm", lambda avg: m.flow_matching_loss(x, cond, average=avg, t=cu(g["fm_t"]), noise=cu(g["fm_eps"]))))
    assert cases
    for pfx, call in cases:
        m.zero_grad(set_to_none=True)
        loss = call(True)
        want = float(g[pfx + "_loss"])
        assert abs(float(loss) - want) <= 1e-5 * abs(want) + 1e-7, (pfx, float(loss), want)
        loss.backward()
        _check_grads(m, g, pfx)
        rows = call(False)
        close(rows, g[pfx + "_loss_rows"], tol=2e-5, what=pfx + " rows", rel=True)


def test_flat_gradient_buffer_path_matches_goldens_and_survives_zero_grad():
    """distributed.attach_flat_grads: the fused loss accumulates straight into ONE flat buffer that every .grad
    aliases; values equal the goldens, two backward calls accumulate, zero_grad(set_to_none=True) is survived,
    and under torch.no_grad() no gradient work is done at all."""
    from generativerl_preview_amd.distributed import attach_flat_grads
    name = "small_vp_noise"
    kind, cfg, m = um.build(name, DEV)
    g = gu.load("model_%s.npz" % name)
    x, cond = cu(g["x"]), cu(g["cond"])
    fg = attach_flat_grads(m.model)
    call = lambda: m.score_matching_loss(x, cond, weighting_scheme="vanilla", t=cu(g["sm_t"]), noise=cu(g["sm_eps"]))  # noqa: E731
    call().backward()
    for k, p in m.named_parameters():
        if k.startswith("model.") and p.requires_grad:
            assert p.grad.data_ptr() >= fg.flat.data_ptr() and p.grad.data_ptr() < fg.flat.data_ptr() + 4 * fg.n, k
    _check_grads(m, g, "sm")
    once = fg.grads.clone()
    call().backward()                                     # accumulates like autograd does
    close(fg.grads, 2 * once, tol=1e-6, what="flat grads accumulate", rel=True)
    torch.optim.SGD(m.parameters(), lr=0.1).zero_grad(set_to_none=True)
    call().backward()                                     # views re-installed on a zeroed buffer
    _check_grads(m, g, "sm")
    fg.zero()
    with torch.no_grad():
        loss = call()
    assert not loss.requires_grad and float(fg.grads.abs().max()) == 0.0
    assert abs(float(loss) - float(g["sm_loss"])) <= 1e-5 * abs(float(g["sm_loss"]))


def test_weighted_rows_backward_and_trainable_fourier():
    """average=False + per-row weights (the GMPO weighted-loss pattern) and a trainable Fourier W
    (swiss-roll tutorial) against torch autograd on the CPU oracle."""
    name = "small_vp_noise"
    kind, cfg, m = um.build(name, DEV)
    m.model.model["t_encoder"].W.requires_grad_(True)
    g = gu.load("model_%s.npz" % name)
    x, cond, t, eps = cu(g["x"]), cu(g["cond"]), cu(g["sm_t"]), cu(g["sm_eps"])
    w = torch.linspace(0.1, 2.0, x.shape[0])
    rows = m.score_matching_loss(x, cond, weighting_scheme="vanilla", average=False, t=t, noise=eps)
    (rows * w.to(DEV)).sum().backward()
    sd = gu.weights(name)
    keys = [k for k in sd if k.startswith("model.")]
    for k in keys:
        sd[k].requires_grad_(True)
    fn = lambda tt, y: O.intrinsic_forward(sd, tt, y, gu.T(g["cond"]))  # noqa: E731
    ref = O.score_matching_loss(cfg["model"]["type"], O.Path(cfg["path"]), fn, gu.T(g["x"]), gu.T(g["sm_t"]),
                                gu.T(g["sm_eps"]), average=False)
    grads = dict(zip(keys, torch.autograd.grad((ref * w).sum(), [sd[k] for k in keys])))
    for k, p in m.named_parameters():
        if k.startswith("model."):
            close(p.grad, grads[k], tol=3e-4, what=k, rel=True)


def test_torch_compile_wrapped_model_is_usable():
    """The reference's algorithms wrap their policies in torch.compile (grl/algorithms/qgpo.py:350, base.py:130; its own
    test grl/unittest/generative_models/diffusion_model/test_diffusion_model.py:86-88 calls the loss and sample methods of
    the wrapper).  The wrapper forwards attribute access to the module, so the HIP path is reached unchanged."""
    kind, cfg, m = um.build("small_vp_noise", DEV)
    cm = torch.compile(m)
    g = gu.load("model_small_vp_noise.npz")
    x, cond = cu(g["x"]), cu(g["cond"])
    a = m.score_matching_loss(x, cond, weighting_scheme="vanilla", t=cu(g["sm_t"]), noise=cu(g["sm_eps"]))
    b = cm.score_matching_loss(x, cond, weighting_scheme="vanilla", t=cu(g["sm_t"]), noise=cu(g["sm_eps"]))
    assert torch.equal(a.detach(), b.detach())
    b.backward()
    assert all(p.grad is not None for k, p in m.named_parameters() if k.startswith("model.") and not k.endswith("t_encoder.W"))
    ts = torch.linspace(0, 1, 6)
    assert torch.equal(m.sample(t_span=ts, x_0=x[:8], condition=cond[:8]), cm.sample(t_span=ts, x_0=x[:8], condition=cond[:8]))
    assert set(cm.state_dict()) == set("_orig_mod." + k for k in m.state_dict()) or set(cm.state_dict()) == set(m.state_dict())


def test_gradients_are_bit_reproducible():
    """Two backward passes over the same batch (20 000 rows: many row slices per dW launch, trainable Fourier W) give
    bit-identical gradients: the slices' partial sums are added in a fixed order, no float atomics."""
    name = "icfm_cond"
    kind, cfg, m = um.build(name, DEV)
    m.model.model["t_encoder"].W.requires_grad_(True)
    gen = torch.Generator().manual_seed(5)
    B = 20000
    x0, x1 = torch.randn(B, 6, generator=gen).to(DEV), torch.rand(B, 6, generator=gen).to(DEV)
    cond, t, eps = torch.randn(B, 17, generator=gen).to(DEV), torch.rand(B, generator=gen).to(DEV), torch.randn(B, 6, generator=gen).to(DEV)
    runs = []
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        m.flow_matching_loss(x0, x1, cond, t=t, noise=eps).backward()
        runs.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    assert len(runs[0]) > 10
    for k in runs[0]:
        assert torch.equal(runs[0][k], runs[1][k]), "gradient of %s differs between two identical passes" % k


def test_two_piece_backward_split_is_opt_in_close_and_reproducible():
    """grlhip_plan_set_backward_split(2): the backward GEMMs form each fp32 product from two bf16 pieces per value (three
    MFMAs) instead of three (six).  Default stays 3; the loss (forward) is untouched bit for bit; the gradients move by
    ~1e-5 of their norm, stay inside the golden gate on the reference's vectors, and stay bit-reproducible."""
    name = "icfm_cond"
    kind, cfg, m = um.build(name, DEV)
    assert m.model.plan.backward_split == 3
    gen = torch.Generator().manual_seed(11)
    B = 20000
    x0, x1 = torch.randn(B, 6, generator=gen).to(DEV), torch.rand(B, 6, generator=gen).to(DEV)
    cond, t, eps = torch.randn(B, 17, generator=gen).to(DEV), torch.rand(B, generator=gen).to(DEV), torch.randn(B, 6, generator=gen).to(DEV)

    def grads():
        m.zero_grad(set_to_none=True)
        loss = m.flow_matching_loss(x0, x1, cond, t=t, noise=eps)
        loss.backward()
        return loss.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    l3, g3 = grads()
    m.model.backward_split = 2
    l2, g2 = grads()
    assert m.model.plan.backward_split == 2
    l2b, g2b = grads()
    assert torch.equal(l3, l2) and torch.equal(l2, l2b)
    differs = 0
    for k in g3:
        assert torch.equal(g2[k], g2b[k]), "two-piece gradient of %s differs between two identical passes" % k
        err = float((g2[k] - g3[k]).double().norm() / g3[k].double().norm().clamp_min(1e-30))
        assert err <= 1e-4, (k, err)
        differs += int(not torch.equal(g2[k], g3[k]))
    assert differs > 10, "the option changed nothing: is the two-piece kernel running?"
    g = gu.load("model_%s.npz" % name)
    m.zero_grad(set_to_none=True)
    m.flow_matching_loss(cu(g["eps"]), cu(g["x"]), cu(g.get("cond")), t=cu(g["fm_t"]), noise=cu(g["fm_eps"])).backward()
    _check_grads(m, g, "fm")
    m.model.backward_split = 3
    l3b, g3b = grads()
    assert all(torch.equal(g3[k], g3b[k]) for k in g3)
    with pytest.raises(L.GrlHipError):
        m.model.plan.set_backward_split(4)


def test_training_batch_spanning_two_chunks_vs_oracle():
    """The batch engine walks the rows in chunks of 65 536: per-row losses and every gradient of a batch
    that crosses the boundary (and is not a multiple of the 128-row tile) against torch autograd on the oracle."""
    name = "small_icfm"
    kind, cfg, m = um.build(name, DEV)
    B = 65536 + 333
    gen = torch.Generator().manual_seed(5)
    x0, x1 = torch.randn(B, 6, generator=gen), torch.rand(B, 6, generator=gen) * 2 - 1
    cond, t, eps = torch.randn(B, 17, generator=gen), torch.rand(B, generator=gen), torch.randn(B, 6, generator=gen)
    rows = m.flow_matching_loss(x0.to(DEV), x1.to(DEV), cond.to(DEV), average=False, t=t.to(DEV), noise=eps.to(DEV))
    rows.mean().backward()
    sd = gu.weights(name)
    keys = [k for k in sd if k.startswith("model.") and not k.endswith("t_encoder.W")]
    for k in keys:
        sd[k].requires_grad_(True)
    ref = O.icfm_flow_matching_loss(lambda tt, y: O.intrinsic_forward(sd, tt, y, cond), x0, x1, t, eps,
                                    cfg["path"]["sigma"], average=False)
    close(rows, ref.detach(), tol=2e-5, what="loss rows", rel=True)
    grads = dict(zip(keys, torch.autograd.grad(ref.mean(), [sd[k] for k in keys])))
    n = 0
    for k, p in m.named_parameters():
        if k in grads:
            close(p.grad, grads[k], tol=3e-4, what=k, rel=True)
            n += 1
    assert n > 10


def test_sampling_rows_spanning_two_chunks_vs_oracle():
    """fp32 sampling walks the rows in chunks of 262 144."""
    name = "small_vp_noise"
    kind, cfg, m = um.build(name, DEV)
    R = 262144 + 77
    gen = torch.Generator().manual_seed(6)
    x0, cond = torch.randn(R, 6, generator=gen), torch.randn(R, 17, generator=gen)
    t_span = torch.linspace(0.0, 1.0, 4)
    traj = m.sample_forward_process(t_span=t_span, x_0=x0.to(DEV), condition=cond.to(DEV)).cpu()
    sd = gu.weights(name)
    for sl in (slice(0, 200), slice(262144 - 100, 262144 + 77)):
        want = O.sample_forward_process(sd, cfg, "diffusion", t_span, x_0=x0[sl], condition=cond[sl])
        close(traj[:, sl], want, what="trajectory rows %s" % (sl,))


def test_adam_step_moves_the_loss_down():
    kind, cfg, m = um.build("small_icfm", DEV)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    gen = torch.Generator().manual_seed(3)
    x1 = torch.rand(256, 6, generator=gen).mul(2).sub(1).to(DEV)
    cond = torch.randn(256, 17, generator=gen).to(DEV)
    x0, t, eps = torch.randn(256, 6, generator=gen).to(DEV), torch.rand(256, generator=gen).to(DEV), torch.randn(256, 6, generator=gen).to(DEV)
    losses = []
    for _ in range(20):
        opt.zero_grad()
        loss = m.flow_matching_loss(x0, x1, cond, t=t, noise=eps)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < 0.9 * losses[0] and all(b < a for a, b in zip(losses, losses[1:])), losses


@pytest.mark.parametrize("dt,Tn", [(0.35, 11), (0.01, 101), (0.03, 11)])
def test_sde_dt_not_matching_the_grid_vs_oracle(dt, Tn):
    """torchsde semantics when dt != t_span spacing: several outputs inside one step (two emits +
    host interpolation), interpolated outputs, and the fp32 round-off case of the C4 config."""
    name = "small_uncond_noise"
    kind, cfg, m = um.build(name, DEV)
    sd = gu.weights(name)
    t_span = torch.linspace(0.0, 1.0, Tn)
    from generativerl_preview_amd.numerical_methods.numerical_solvers import SDESolver
    n_inner = len(SDESolver(dt=dt)._schedule(t_span)[0])
    gen = torch.Generator().manual_seed(8)
    rows = 40
    xT = torch.randn(rows, 3, generator=gen)
    dW = torch.randn(n_inner, rows, 3, generator=gen) * (dt ** 0.5)
    um.fix_noise(m, xT.to(DEV))
    got = m.sample_forward_process(t_span=t_span, batch_size=rows, noise_increments=dW.to(DEV),
                                   solver_config=dict(type="SDESolver", args=dict(dt=dt)))
    want = O.sample_forward_process(sd, cfg, kind, t_span, batch_size=rows, x_T=xT, solver="SDESolver", dt=dt,
                                    increments=dW)
    close(got, want, tol=2e-4, what="sde dt=%g" % dt)


FIXED_X_MODELS = ("small_vp_noise", "small_vp_score", "small_gvp_velocity", "small_uncond_noise", "small_energy", "qgpo_noise")


@pytest.mark.parametrize("name", FIXED_X_MODELS)
def test_sampling_with_fixed_x_vs_golden(name):
    """sample_forward_process_with_fixed_x / sample_with_fixed_x (grlhip_sample_fixed) against the reference's own
    trajectories (tests/golden/make_golden_fixed_x.py): pinned, free and fractionally masked components, x_0 given or
    drawn, extra batch sizes, guided and unguided, ODE and Euler-Maruyama."""
    kind, cfg, m = um.build(name, DEV)
    g = gu.load("fixed_x.npz")
    pre = name + "/"
    x0, fx, mask = cu(g[pre + "x0"]), cu(g[pre + "fixed_x"]), cu(g[pre + "mask"])
    cond = cu(g[pre + "cond"]) if pre + "cond" in g else None
    t_span = torch.linspace(0.0, 1.0, 7)
    rel = name in REL_TRAJ
    n = 0
    for tag, B, use_x0 in (("x0", None, True), ("B2_x0", 2, True), ("B3_draw", 3, False)):
        for gs in ((0.0, 1.0) if kind == "energy" else (None,)):
            key = "%sode_%s%s" % (pre, tag, "" if gs is None else "_g%g" % gs)
            kw = dict(fixed_x=fx, fixed_mask=mask, t_span=t_span, batch_size=B, x_0=x0 if use_x0 else None,
                      condition=cond, fixed_noise=cu(g["%sode_%s_noise" % (pre, tag)]))
            if gs is not None:
                kw["guidance_scale"] = gs
            if not use_x0:
                um.fix_noise(m, cu(g["%sode_%s_xT" % (pre, tag)]))
            traj = m.sample_forward_process_with_fixed_x(**kw)
            close(traj, g[key], what=key, rel=rel)
            final = m.sample_with_fixed_x(**kw)
            close(final, g[key][-1], what=key + " final", rel=rel)
            n += 1
    for gs in ((0.0, 1.0) if kind == "energy" else (None,)):
        key = "%ssde%s" % (pre, "" if gs is None else "_g%g" % gs)
        if key not in g:
            continue
        kw = dict(fixed_x=fx, fixed_mask=mask, t_span=torch.linspace(0.0, 1.0, 5), x_0=x0, condition=cond,
                  fixed_noise=cu(g[pre + "sde_noise"]), noise_increments=cu(g[pre + "sde_dW"]),
                  solver_config=dict(type="SDESolver", args=dict(library="torchsde", dt=0.25)))
        if gs is not None:
            kw["guidance_scale"] = gs
        close(m.sample_forward_process_with_fixed_x(**kw), g[key], what=key)
        n += 1
    assert n >= 3


def test_sampling_with_fixed_x_shapes_and_invariants():
    """The calls of the reference's own unit test (grl/unittest/generative_models/diffusion_model/
    test_diffusion_model.py:158-233) on a swiss-roll-sized model, plus what the arithmetic implies: pinned components
    (mask 0) come back as fixed_x at every time, rows with nothing pinned follow the ordinary sampler exactly, a
    263 000-row call spans two chunks of the batch engine, and the unsupported combinations raise."""
    kind, cfg, m = um.build("swiss_noise", DEV)
    rows = 12
    data = torch.randn(rows, 2, device=DEV)
    fixed_x = torch.tensor([1.0, 0.0], device=DEV).unsqueeze(0).expand(rows, -1)
    fixed_mask = torch.tensor([0.0, 1.0], device=DEV).unsqueeze(0).expand(rows, -1)
    t_span = torch.linspace(0.0, 1.0, 5)
    f = m.sample_forward_process_with_fixed_x
    assert tuple(f(fixed_x=fixed_x, fixed_mask=fixed_mask, t_span=t_span).shape) == (5, rows, 2)
    assert tuple(f(fixed_x=fixed_x, fixed_mask=fixed_mask, t_span=t_span, batch_size=20).shape) == (5, 20, rows, 2)
    assert tuple(f(fixed_x=fixed_x, fixed_mask=fixed_mask, t_span=t_span, x_0=data).shape) == (5, rows, 2)
    x_t = f(fixed_x=fixed_x, fixed_mask=fixed_mask, t_span=t_span, batch_size=20, x_0=data)
    assert tuple(x_t.shape) == (5, 20, rows, 2)
    assert torch.equal(x_t[..., 0], torch.ones_like(x_t[..., 0]))          # the pinned component, at every time
    s = m.sample_with_fixed_x
    assert tuple(s(fixed_x=fixed_x, fixed_mask=fixed_mask, t_span=t_span).shape) == (rows, 2)
    assert tuple(s(fixed_x=fixed_x, fixed_mask=fixed_mask, t_span=t_span, batch_size=20).shape) == (20, rows, 2)
    assert tuple(s(fixed_x=fixed_x, fixed_mask=fixed_mask, t_span=t_span, x_0=data).shape) == (rows, 2)
    assert tuple(s(fixed_x=fixed_x, fixed_mask=fixed_mask, t_span=t_span, batch_size=20, x_0=data).shape) == (20, rows, 2)
    # nothing pinned: the ordinary sampler (the fp32 batch engine runs both)
    ones = torch.ones(rows, 2, device=DEV)
    free = f(fixed_x=fixed_x, fixed_mask=ones, t_span=t_span, x_0=data)
    close(free, m.sample_forward_process(t_span=t_span, x_0=data), what="fixed_x with an all-ones mask", tol=1e-6)
    # two chunks of the batch engine, against the oracle on the rows around the chunk boundary
    big = 263000   # SAMPLE_CHUNK_ROWS = 262 144
    gen = torch.Generator(device="cpu").manual_seed(5)
    x0 = torch.randn(big, 2, generator=gen)
    fxb = torch.randn(big, 2, generator=gen)
    mk = (torch.rand(big, 2, generator=gen) < 0.5).float()
    noise = torch.randn(4, big, 2, generator=gen)
    out = s(fixed_x=cu(fxb), fixed_mask=cu(mk), t_span=t_span, x_0=cu(x0), fixed_noise=cu(noise))
    sel = torch.cat([torch.arange(0, 40), torch.arange(262144 - 20, 262144 + 20), torch.arange(big - 40, big)])
    ref = O.sample_forward_process_with_fixed_x(gu.weights("swiss_noise"), cfg, kind, t_span, fxb[sel], mk[sel],
                                                noise[:, sel], x_0=x0[sel])[-1]
    close(out[sel.to(DEV)], ref, what="fixed_x 263000 rows (two chunks)")
    with pytest.raises(NotImplementedError):
        s(fixed_x=fixed_x, fixed_mask=fixed_mask, t_span=t_span, with_grad=True)
    with pytest.raises(AssertionError):
        s(fixed_x=fixed_x, fixed_mask=fixed_mask, solver_config=dict(type="DPMSolver", args=dict(
            order=2, device=DEV, steps=5)))
    with pytest.raises(AssertionError):
        s(fixed_x=fixed_x, fixed_mask=fixed_mask[:3], t_span=t_span)


FORWARD_MODELS = ("small_vp_noise", "small_vp_score", "small_vp_velocity", "small_vp_data", "small_gvp_velocity",
                  "small_uncond_noise", "small_icfm", "swiss_noise", "icfm_cond")


@pytest.mark.parametrize("name", FORWARD_MODELS)
def test_forward_sample_vs_golden(name):
    """forward_sample / forward_sample_process (data -> latent; the fused integrator with the coefficients of
    DiffusionProcess.forward_ode / forward_sde, or the ICFM's reversed flow) against the reference's trajectories."""
    kind, cfg, m = um.build(name, DEV)
    g = gu.load("forward.npz")
    pre = name + "/"
    x = cu(g[pre + "x"])
    cond = cu(g[pre + "cond"]) if pre + "cond" in g else None
    t_span = torch.linspace(0.05, 1.0, 8)
    traj = m.forward_sample_process(x=x, t_span=t_span, condition=cond)
    close(traj, g[pre + "ode"], what=pre + "ode", rel=True)
    close(m.forward_sample(x=x, t_span=t_span, condition=cond), g[pre + "ode"][-1], what=pre + "ode final", rel=True)
    if pre + "sde" in g:
        traj = m.forward_sample_process(x=x, t_span=torch.tensor([0.05, 0.3, 0.55, 0.8, 1.0]), condition=cond,
                                        noise_increments=cu(g[pre + "sde_dW"]),
                                        solver_config=dict(type="SDESolver", args=dict(library="torchsde", dt=0.25)))
        close(traj, g[pre + "sde"], what=pre + "sde", rel=True)
    if name in ("swiss_noise", "icfm_cond"):   # [512,256,128]: the persistent fused bf16 integrator, other coefficients
        m.model.compute_dtype = L.BF16
        close(m.forward_sample_process(x=x, t_span=t_span, condition=cond), g[pre + "ode"], tol=2e-2,
              what=pre + "ode bf16", rel=True)
        assert m.model.plan.fused_kernel()


def test_forward_sample_then_sample_round_trip():
    """Encoding then decoding with the same model and grid returns close to the data (Euler both ways, 200 steps): the
    two directions use one network and mirrored coefficients."""
    kind, cfg, m = um.build("icfm_cond", DEV)
    g = gu.load("forward.npz")
    x, cond = cu(g["icfm_cond/x"]), cu(g["icfm_cond/cond"])
    t_span = torch.linspace(0.0, 1.0, 201)
    z = m.forward_sample(x=x, t_span=t_span, condition=cond)
    back = m.sample(t_span=t_span, x_0=z, condition=cond)
    close(back, x, tol=1e-3, what="icfm encode -> decode round trip")


@pytest.mark.parametrize("name", ["small_icfm", "icfm_cond"])
def test_icfm_masked_loss_and_sampling_vs_golden(name):
    """IndependentConditionalFlowModel.flow_matching_loss_with_mask (loss, per-row losses, every parameter gradient --
    the differentiable HIP forward behind autograd), the element-wise loss under it, and sample_with_mask, against
    the reference (grl/algorithms/gmpg_mask.py is their caller)."""
    kind, cfg, m = um.build(name, DEV)
    g = gu.load("icfm_mask.npz")
    pre = name + "/"
    x0, x1, cond, mask = (cu(g[pre + k]) for k in ("x0", "x1", "cond", "mask"))
    t, eps = cu(g[pre + "t"]), cu(g[pre + "eps"])
    m.zero_grad(set_to_none=True)
    loss = m.flow_matching_loss_with_mask(x0, x1, cond, mask, average=True, t=t, noise=eps)
    want = float(g[pre + "loss"])
    assert abs(float(loss) - want) <= 1e-5 * abs(want) + 1e-7, (float(loss), want)
    loss.backward()
    n = 0
    for k, p in m.named_parameters():
        if pre + "grad/" + k in g:
            close(p.grad, g[pre + "grad/" + k], tol=2e-4, what="masked loss grad " + k, rel=True)
            n += 1
        elif pre + "gradsum/" + k in g and not k.endswith("t_encoder.W"):
            s = g[pre + "gradsum/" + k]
            assert abs(float(p.grad.double().pow(2).sum().sqrt()) - s[1]) <= 2e-4 * s[1] + 1e-7, k
            n += 1
    assert n > 10
    close(m.flow_matching_loss_with_mask(x0, x1, cond, mask, average=False, t=t, noise=eps), g[pre + "loss_rows"],
          tol=2e-5, what="masked loss rows", rel=True)
    close(m.flow_matching_loss(x0, x1, cond, average=False, sum_all_elements=False, t=t, noise=eps),
          g[pre + "loss_elements"], tol=2e-5, what="element-wise loss", rel=True)
    close(m.flow_matching_loss_with_mask(x0, x1, cond, None, t=t, noise=eps),
          m.flow_matching_loss(x0, x1, cond, t=t, noise=eps), tol=0.0, what="mask=None")
    t_span = torch.linspace(0.0, 1.0, 6)
    traj = m.sample_with_mask_forward_process(t_span=t_span, x_0=x0, condition=cond, mask=mask, x_1=x1)
    close(traj, g[pre + "sample_traj"], what="sample_with_mask_forward_process")
    close(m.sample_with_mask(t_span=t_span, x_0=x0, condition=cond, mask=mask, x_1=x1), g[pre + "sample_traj"][-1],
          what="sample_with_mask")


LOGPROB_MODELS = ("small_vp_noise", "small_vp_score", "small_gvp_velocity", "small_uncond_noise", "small_icfm")


@pytest.mark.parametrize("name", LOGPROB_MODELS)
def test_log_prob_vs_golden(name):
    """log_prob of the diffusion models (metric.compute_likelihood) and of the ICFM: network forwards and the
    vector-Jacobian products of the divergence are library calls; exact trace and Hutchinson estimator (replayed
    probes) against the reference, relative 1e-3 (the sums run over 40 steps of values up to 1e3)."""
    from generativerl_preview_amd.generative_models.metric import compute_likelihood
    kind, cfg, m = um.build(name, DEV)
    g = gu.load("logprob.npz")
    pre = name + "/"
    x, grid = cu(g[pre + "x"]), cu(g[pre + "grid"])
    cond = cu(g[pre + "cond"]) if pre + "cond" in g else None
    if kind == "icfm":
        exact = m.log_prob(x_1=x, condition=cond, t=grid, using_Hutchinson_trace_estimator=False)
        hutch = m.log_prob(x_1=x, condition=cond, t=grid, noise=cu(g[pre + "probes"]))
    else:
        exact = compute_likelihood(m, x, t=grid, condition=cond, using_Hutchinson_trace_estimator=False)
        hutch = m.log_prob(x=x, condition=cond, t=grid, noise=cu(g[pre + "probes"]))
    close(exact, g[pre + "exact"], tol=1e-3, what=pre + "log_prob exact trace", rel=True)
    close(hutch, g[pre + "hutchinson"], tol=1e-3, what=pre + "log_prob hutchinson", rel=True)
    if pre + "log_prob_default" in g:   # DiffusionModel.log_prob's own grid: 999 Euler steps
        close(m.log_prob(x=x, noise=cu(g[pre + "probes_default"])), g[pre + "log_prob_default"], tol=1e-3,
              what=pre + "log_prob default grid", rel=True)
        xs, lp = m.sample_with_log_prob(t_span=torch.linspace(0.0, 1.0, 11), batch_size=4)
        assert tuple(xs.shape) == (4, cfg["x_size"]) and tuple(lp.shape) == (4,) and bool(torch.isfinite(lp).all())
        # with_grad=True: the sample carries a grad_fn, the log-probability does not (and a warning says so)
        with pytest.warns(UserWarning, match="without gradient"):
            xs, lp = m.sample_with_log_prob(t_span=torch.linspace(0.0, 1.0, 11), batch_size=4, with_grad=True)
        assert xs.requires_grad and not lp.requires_grad and bool(torch.isfinite(lp).all())
    if kind == "icfm":
        xs, lp = m.sample_with_log_prob(t_span=torch.linspace(0.0, 1.0, 11), x_0=x, condition=cond)
        assert tuple(xs.shape) == tuple(x.shape) and tuple(lp.shape) == (x.shape[0],) and bool(torch.isfinite(lp).all())
        with_prior = m.log_prob(x_1=x, condition=cond, t=grid, using_Hutchinson_trace_estimator=False,
                                log_prob_x_0=torch.zeros(x.shape[0], device=DEV))
        ones = m.log_prob(x_1=x, condition=cond, t=grid, using_Hutchinson_trace_estimator=False,
                          function_log_prob_x_0=lambda z: torch.ones(z.shape[0], device=DEV))
        close(ones - with_prior, torch.ones(x.shape[0]), tol=1e-5, what="prior log-density options")


def test_dpo_loss_vs_golden():
    """DiffusionModel.dpo_loss against the reference (loss 1e-5 relative, every parameter gradient): the four network
    evaluations are the differentiable HIP forward."""
    kind, cfg, m = um.build("swiss_noise", DEV)
    _, _, ref = um.build("swiss_noise", DEV)
    g = gu.load("dpo.npz")
    w = gu.weights("swiss_noise", as_torch=False)
    gen = torch.Generator().manual_seed(4242)   # tests/golden/make_golden_dpo.py: the frozen model's perturbation
    sd = {k: torch.from_numpy(np.asarray(v)) + 0.02 * torch.randn(np.asarray(v).shape, generator=gen) * (0.0 if k.endswith("t_encoder.W") else 1.0)
          for k, v in w.items()}
    assert abs(sum(float(v.double().sum()) for v in sd.values()) - float(g["ref_checksum"])) < 1e-6 * abs(float(g["ref_checksum"])) + 1e-6
    ref.load_state_dict(sd)
    ref = ref.to(DEV)
    m.zero_grad(set_to_none=True)
    loss = m.dpo_loss(ref, cu(g["data"]), float(g["beta"]), t=cu(g["t"]), noise=cu(g["noise"]))
    want = float(g["loss"])
    assert abs(float(loss) - want) <= 1e-5 * abs(want) + 1e-7, (float(loss), want)
    loss.backward()
    n = 0
    for k, p in m.named_parameters():
        if "gradsub/" + k in g and not k.endswith("t_encoder.W"):
            flat = p.grad.reshape(-1)
            close(flat[:: max(1, flat.numel() // 509)][:509], g["gradsub/" + k], tol=2e-4, what="dpo grad " + k, rel=True)
            s = g["gradsum/" + k]
            assert abs(float(flat.double().pow(2).sum().sqrt()) - s[1]) <= 2e-4 * s[1] + 1e-7, k
            n += 1
    assert n > 10
    assert all(p.grad is None for p in ref.parameters())


def test_sample_fixed_c_abi_dtypes_and_edges():
    """grlhip_sample_fixed through NetPlan.sample_fixed: GRLHIP_BF16 on a net without a fused integrator (batch engine,
    bf16 MFMAs) within the bf16 tolerance of the oracle; a net whose bf16 image belongs to the fused integrator is
    refused; zero steps return the blended initial state; zero rows are a no-op; a null mask is an argument error."""
    from generativerl_preview_amd.generative_models.fused import FusedDrift
    kind, cfg, m = um.build("small_gvp_velocity", DEV)
    g = gu.load("fixed_x.npz")
    pre = "small_gvp_velocity/"
    x0, fx, mask, cond = cu(g[pre + "x0"]), cu(g[pre + "fixed_x"]), cu(g[pre + "mask"]), cu(g[pre + "cond"])
    noise = cu(g[pre + "ode_x0_noise"])
    t_span = torch.linspace(0.0, 1.0, 7)
    net = m.model
    drift = FusedDrift("reverse_ode", net, m.model_type, m.path, condition=cond)
    ts = t_span.to("cpu", torch.float32)
    table = drift.table(ts[:-1], ts[1:] - ts[:-1], torch.arange(1, 7), torch.ones(6))
    tau = table[:, 0]
    fix_sc = torch.stack([m.path.scale(tau), m.path.std(tau)], 1).float().contiguous()
    assert net.plan.fused_kernel() == ""
    final, traj = net.plan.sample_fixed(net.packed(L.BF16), x0, cond, table, fx, mask, fix_sc, fix_noise=noise,
                                        trajectory_slots=7, dtype=L.BF16)
    close(traj, g[pre + "ode_x0"], tol=BF16_TOL, what="sample_fixed bf16 batch engine")
    exact, _ = net.plan.sample_fixed(net.packed(L.F32), x0, cond, table, fx, mask, fix_sc, fix_noise=noise)
    assert float((final - exact).abs().max()) > 0.0
    # zero steps: the blended initial state
    zf, zt = net.plan.sample_fixed(net.packed(L.F32), x0, cond, table[:0], fx, mask, fix_sc[:0], trajectory_slots=1)
    close(zf, fx * (1 - mask) + x0 * mask, tol=0.0, what="sample_fixed zero steps")
    close(zt[0], zf, tol=0.0, what="sample_fixed zero steps slot 0")
    # zero rows
    ef, _ = net.plan.sample_fixed(net.packed(L.F32), x0[:0], cond[:0], table, fx[:0], mask[:0], fix_sc)
    assert tuple(ef.shape) == (0, 6)
    # the fused integrator's image is refused
    kind2, cfg2, m2 = um.build("swiss_noise", DEV)
    n2 = m2.model
    assert n2.plan.fused_kernel()
    d2 = FusedDrift("reverse_ode", n2, m2.model_type, m2.path)
    tab2 = d2.table(ts[:-1], ts[1:] - ts[:-1], torch.arange(1, 7), torch.ones(6))
    z = torch.zeros(4, 2, device=DEV)
    with pytest.raises(L.GrlHipError, match="fused"):
        n2.plan.sample_fixed(n2.packed(L.BF16), z, None, tab2, z, z, fix_sc, dtype=L.BF16)
    # null mask
    rc = L.lib().grlhip_sample_fixed(n2.plan.handle, n2.packed(L.F32).data_ptr(), L.F32, None, None, z.data_ptr(), None,
                                     tab2.to(DEV).data_ptr(), None, 6, 4, z.data_ptr(), None, fix_sc.to(DEV).data_ptr(), None,
                                     z.data_ptr(), None, None, 0, None)
    assert rc == -1   # GRLHIP_ERR_INVALID
