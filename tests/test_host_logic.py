"""CPU tests: C-ABI exports, plan/param layout against the reference state_dict manifest, host-side
schedules and coefficient tables, solver step schedules, no-fallback behaviour."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import golden_lib as gl
import golden_util as gu
import util_models as um
from generativerl_preview_amd import _lib as L
from generativerl_preview_amd.generative_models.fused import FusedDrift
from generativerl_preview_amd.numerical_methods.monte_carlo import MonteCarloSampler
from generativerl_preview_amd.numerical_methods.numerical_solvers import SDESolver, get_solver
from generativerl_preview_amd.numerical_methods.probability_path import GaussianConditionalProbabilityPath
from oracle import grl_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "grlhip.h")).read()
    declared = set(re.findall(r"\b(grlhip_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(L.PROTOTYPES), declared ^ set(L.PROTOTYPES)
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in declared:
        assert getattr(lib, name) is not None


@pytest.mark.parametrize("name", list(gl.model_zoo()))
def test_state_dict_matches_reference_keys(name):
    kind, cfg, m = um.build(name)
    want = [(k, tuple(s)) for k, s in gu.MANIFEST["models"][name]["key_shapes"]]
    got = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert got == want
    # the plan consumes exactly the IntrinsicModel's parameters
    names = [n for n, _ in m.model.plan.params]
    assert sorted(names) == sorted(k for k, _ in m.model.model.named_parameters())


@pytest.mark.parametrize("name,blocks_with_modify,blocks,bwd_launches", [("qgpo_noise", 3, 6, 28), ("gvp_wide", 1, 2, 14),
                                                                          ("swiss_noise", 3, 6, 26)])
def test_batch_engine_fuses_time_mlp_and_modify_x(name, blocks_with_modify, blocks, bwd_launches):
    """The peephole of the batch engine: every residual block's time-MLP, and modify_x where the block changes
    width, become the second accumulator of the consuming GEMM; nothing in the backward program may fuse."""
    kind, cfg, m = um.build(name)
    h = m.model.plan.handle
    info = {}
    for prog in range(4):
        n_ops, n_launch = ctypes.c_int32(), ctypes.c_int32()
        L.check(L.lib().grlhip_plan_program_info(h, prog, ctypes.byref(n_ops), ctypes.byref(n_launch)))
        info[prog] = (n_ops.value, n_launch.value)
    fused = blocks + blocks_with_modify
    assert info[0][0] - info[0][1] == fused, info
    assert info[2][0] - info[2][1] == fused, info
    # backward: no dual GEMMs; the weight gradients wait until one of their operands is about to be overwritten and then
    # go as ONE launch (a residual block's three or four together: 8 launches instead of 25 for the [512,256,128] net),
    # every one writes per-slice partials, and ONE launch per 24 of them adds the slices in order at the end
    n_dw = sum(1 for n, _ in m.model.plan.params if n.endswith(".weight"))
    reduces = (n_dw + 23) // 24
    # ... and two input-gradient GEMMs of a block that do not depend on each other (time-MLP beside dense1, modify_x
    # beside dense2) go as one launch (k_bt_gemm_x6_pair)
    assert info[3][0] > 0 and info[3][1] == bwd_launches, (info, n_dw)
    assert info[3][1] < info[3][0] - n_dw + (n_dw + 2) // 3 + reduces, (info, n_dw)
    assert info[1] == (0, 0)


def test_energy_net_plan_has_training_programs():
    """ConcatenateMLP [55, 256, 256] -> 1 (CEP loss): 2 forward linears keeping pre-activations + the matrix-vector
    output layer; backward = the output layer's one-pass backward (dZ of the layer below, its weight / bias gradient), 2
    weight gradients, the transposed GEMM whose epilogue applies SiLU', the GEMM that only feeds the Fourier-frequency
    gradient, and the trailing input-gradient GEMM; 7 parameters in the flat gradient."""
    from generativerl_preview_amd.engine import NetPlan
    plan = NetPlan(L.NET_CONCAT_MLP, 6, 1, 17, 32, 0, 0, [256, 256])
    info = {}
    for prog in range(4):
        n_ops, n_launch = ctypes.c_int32(), ctypes.c_int32()
        L.check(L.lib().grlhip_plan_program_info(plan.handle, prog, ctypes.byref(n_ops), ctypes.byref(n_launch)))
        info[prog] = (n_ops.value, n_launch.value)
    assert info[2] == (3, 3) and info[3] == (6, 6), info   # 4 ops + ONE launch for the 2 queued weight gradients + the one batched ordered reduce of the 3 gradients
    assert len(plan.params) == 7
    assert L.lib().grlhip_grad_numel(plan.handle) == 16 + 55 * 256 + 256 + 256 * 256 + 256 + 256 + 1
    assert L.lib().grlhip_loss_workspace_bytes(plan.handle, 4096, L.F32) > 0


def test_workspace_size_follows_the_slab_pitch():
    """The slab's row pitch is rows rounded up to 128, and to 256 above 32 768 rows (so that the 256-tile bf16 GEMM covers
    any large batch); batches beyond the chunk size are processed in chunks of 262 144 rows.  Seen from outside: the
    workspace size is a step function of the row count with exactly those steps, and never decreases."""
    from generativerl_preview_amd.engine import NetPlan
    plan = NetPlan(L.NET_RESNET, 64, 64, 0, 32, 0, 0, [128])
    size = lambda rows: L.lib().grlhip_workspace_bytes(plan.handle, None, rows, L.BF16)  # noqa: E731
    assert size(1) == size(128) < size(129) == size(256)
    assert size(32768 - 127) == size(32768) < size(32769) == size(32768 + 256)
    assert size(32768 + 257) == size(32768 + 512) > size(32768 + 256)
    per_row = (size(32768 + 512) - size(32768 + 256)) // 256
    assert per_row > 4 * 128 and (size(256) - size(128)) == 128 * per_row      # bytes per slab row, same below the switch
    assert size(262144) == size(10 ** 7)                                       # chunked beyond 262 144 rows
    prev = 0
    for rows in range(1, 70000, 97):
        cur = size(rows)
        assert cur >= prev
        prev = cur
    # a wide net (hidden >= 256) also has the dense bf16 buffers of the wide engine (256-row pitch at any size): the
    # workspace is the larger of the two layouts -- still a never-decreasing step function, chunked at 262 144 rows
    wide = NetPlan(L.NET_RESNET, 64, 64, 0, 32, 0, 0, [1024])
    wsize = lambda rows: L.lib().grlhip_workspace_bytes(wide.handle, None, rows, L.BF16)  # noqa: E731
    assert wsize(32768 + 1) == wsize(32768 + 256) < wsize(32768 + 257)
    assert wsize(262144) == wsize(10 ** 7) > size(262144)
    prev = 0
    for rows in range(1, 70000, 97):
        cur = wsize(rows)
        assert cur >= prev
        prev = cur


def test_adam_step_validates_its_arguments_without_a_gpu():
    """grlhip_adam_step: argument checks come before any launch (n <= 0 is a no-op, null buffers and step < 1 are
    GRLHIP_ERR_INVALID), and grlhip_plan_set_fourier_grad toggles a plan option on the host."""
    lib = L.lib()
    assert lib.grlhip_adam_step(None, None, None, None, 0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, None) == 0
    assert lib.grlhip_adam_step(None, None, None, None, 16, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, None) != 0
    assert b"null" in lib.grlhip_last_error()
    buf = (ctypes.c_float * 16)()
    ptr = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.grlhip_adam_step(ptr, ptr, ptr, ptr, 16, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, None) != 0
    assert b"step" in lib.grlhip_last_error()
    from generativerl_preview_amd.engine import NetPlan
    plan = NetPlan(L.NET_RESNET, 6, 6, 17, 32, 32, 128, [64, 32, 16])
    assert plan.fourier_grad is True
    plan.set_fourier_grad(False)
    assert plan.fourier_grad is False
    assert lib.grlhip_plan_set_fourier_grad(None, 1) != 0
    # the device-counted variant: a null or misaligned step record is refused before anything is launched
    assert lib.grlhip_adam_step_counted(ptr, ptr, ptr, ptr, 16, 1e-3, 0.9, 0.999, 1e-8, 0.0, None, 1, None) != 0
    assert b"step_record" in lib.grlhip_last_error()
    odd = ctypes.c_void_p(ctypes.addressof(buf) + 4)
    assert lib.grlhip_adam_step_counted(ptr, ptr, ptr, ptr, 16, 1e-3, 0.9, 0.999, 1e-8, 0.0, odd, 0, None) != 0
    # the backward split is 2 or 3
    assert plan.backward_split == 3
    plan.set_backward_split(2)
    assert plan.backward_split == 2
    with pytest.raises(L.GrlHipError):
        plan.set_backward_split(1)
    assert lib.grlhip_plan_set_backward_split(None, 2) != 0


def test_stage_backward_rejects_bad_arguments_without_a_gpu():
    """grlhip_sample_stages_backward / grlhip_sample_ode_backward: the checks that need no device (null plan or weight
    image, wrong plan kind) return an error code and a message instead of launching anything."""
    lib = L.lib()
    from generativerl_preview_amd.engine import NetPlan
    buf = (ctypes.c_float * 64)()
    ptr = ctypes.cast(buf, ctypes.c_void_p)
    for fn in (lib.grlhip_sample_stages_backward, lib.grlhip_sample_ode_backward):
        assert fn(None, ptr, L.F32, ptr, None, ptr, 1, 4, ptr, None, ptr, ptr, 64, None) != 0
        assert b"null plan" in lib.grlhip_last_error()
        plan = NetPlan(L.NET_RESNET, 6, 6, 0, 32, 0, 0, [64, 32, 16])
        assert fn(plan.handle, None, L.F32, ptr, None, ptr, 1, 4, ptr, None, ptr, ptr, 64, None) != 0
        assert b"packed" in lib.grlhip_last_error()


def test_plan_rejects_bad_wiring():
    from generativerl_preview_amd.engine import NetPlan
    with pytest.raises(L.GrlHipError, match="hidden"):
        NetPlan(L.NET_RESNET, 6, 6, 0, 32, 0, 0, [512, 128])
    with pytest.raises(L.GrlHipError):
        NetPlan(L.NET_RESNET, 6, 6, 0, 31, 0, 0, [64])


def test_registry_errors():
    from generativerl_preview_amd.neural_network import get_module
    from generativerl_preview_amd.neural_network.encoders import get_encoder
    with pytest.raises(ValueError):
        get_solver("nope")
    with pytest.raises(ValueError):
        get_module("nope")
    with pytest.raises(ValueError):
        get_encoder("nope")
    assert get_solver("odesolver") is get_solver("ODESolver")


def test_no_cpu_fallback():
    kind, cfg, m = um.build("small_vp_noise")
    x = torch.zeros(4, 6)
    with pytest.raises(L.GrlHipError):
        m.sample(t_span=torch.linspace(0, 1, 5), x_0=x, condition=torch.zeros(4, 17))
    with pytest.raises(L.GrlHipError):
        m.model(torch.zeros(4), x, torch.zeros(4, 17))


def test_schedules_match_golden():
    g = gu.load("schedules.npz")
    t = gu.T(g["t"])
    for pname, pcfg in (("vp", gl.VP), ("gvp", gl.GVP)):
        p = GaussianConditionalProbabilityPath(pcfg)
        for fn in ("drift_coefficient", "diffusion", "diffusion_squared", "scale", "d_scale_dt", "std", "d_std_dt",
                   "covariance"):
            a, b = getattr(p, fn)(t).numpy(), g[pname + "_" + fn]
            fin = np.isfinite(b)
            assert np.array_equal(np.isfinite(a), fin)
            np.testing.assert_allclose(a[fin], b[fin], rtol=1e-6, atol=1e-7)


class _Echo:
    """stands in for an IntrinsicModel when only the coefficient table is exercised"""


@pytest.mark.parametrize("mt", ["score_function", "noise_function", "velocity_function", "data_prediction_function"])
@pytest.mark.parametrize("pcfg", [gl.VP, gl.GVP])
@pytest.mark.parametrize("kind", ["reverse_ode", "reverse_sde"])
def test_coefficient_table_reproduces_the_drift_closures(mt, pcfg, kind):
    """a*x + b*net from the table == the oracle's reverse_ode/sde drift for an arbitrary net output."""
    torch.manual_seed(0)
    p, op = GaussianConditionalProbabilityPath(pcfg), O.Path(pcfg)
    t = torch.linspace(0.05, 0.9, 7)
    x, net = torch.randn(7, 5), torch.randn(7, 5)
    d = FusedDrift(kind, None, mt, p, p)
    tau, a, b, g, s = d.coefficients(t)
    mine = a[:, None] * x + b[:, None] * net
    fn = lambda tt, y: net  # noqa: E731
    ref = O.reverse_ode_drift(mt, op, fn, t, x) if kind == "reverse_ode" else O.reverse_sde_drift(mt, op, op, fn, t, x)
    torch.testing.assert_close(mine, ref, rtol=2e-5, atol=2e-5)
    if kind == "reverse_sde":
        torch.testing.assert_close(s[:, None].expand(7, 5), O.reverse_sde_diffusion(op, t, x))
    # guided form: score head folded in
    dg = FusedDrift(kind, None, mt, p, p, guidance=object(), guidance_scale=2.5)
    tau, a, b, g, s = dg.coefficients(t)
    guid = torch.randn(7, 5)
    mine = a[:, None] * x + b[:, None] * net + g[:, None] * guid
    sfn = lambda tt, y: O.score_head(mt, op, net, tt, y) + 2.5 * guid  # noqa: E731
    ref = (O.reverse_ode_drift("score_function", op, sfn, t, x) if kind == "reverse_ode"
           else O.reverse_sde_drift("score_function", op, op, sfn, t, x))
    torch.testing.assert_close(mine, ref, rtol=5e-5, atol=5e-5)


def test_sde_schedule_matches_torchsde_semantics():
    s = SDESolver(dt=0.03)
    t0, dts, outs = s._schedule(torch.linspace(0, 1, 11))
    assert abs(float(dts.sum()) - 1.0) < 1e-5
    assert [k for k, _, _ in outs] == list(range(1, 11))
    assert all(0 < w <= 1.0 for _, _, w in outs)
    # grid == dt: one step per output, no interpolation
    t0, dts, outs = SDESolver(dt=0.1)._schedule(torch.linspace(0, 1, 11))
    assert len(t0) == 10 and [st for _, st, _ in outs] == list(range(10))
    # dt larger than the spacing: several outputs bracketed by one step -> two emits + host interpolation
    t0, dts, outs = SDESolver(dt=0.35)._schedule(torch.linspace(0, 1, 11))
    emit, lerp, emit2, lerp2, extra = SDESolver._emit_plan(len(t0), outs)
    assert len(t0) == 3 and int((emit2 >= 0).sum()) == 3 and len(extra) == 10 - 6
    # fp32 round-off case of the C4 config (dt = 1/100 on linspace(0, 1, 101)) must tabulate
    t0, dts, outs = SDESolver(dt=0.01)._schedule(torch.linspace(0, 1, 101))
    assert len(outs) == 100


def test_monte_carlo_sampler_matches_golden():
    g = gu.load("monte_carlo.npz")
    path = GaussianConditionalProbabilityPath(gl.VP)
    mc = MonteCarloSampler(lambda t: path.diffusion_squared(t) / (path.covariance(t) + 1e-6), 0.0, 1.0)
    torch.manual_seed(int(g["seed"]))
    np.testing.assert_allclose(mc.sample(4096).numpy(), g["t"], rtol=0, atol=0)


def test_generic_callable_ode_solver():
    s = get_solver("ODESolver")()
    out = s.integrate(lambda t, x: -x, torch.ones(3, 2), torch.linspace(0, 1, 11))
    assert out.shape == (11, 3, 2) and s.nfe == 10
    torch.testing.assert_close(out[-1], torch.full((3, 2), 0.9 ** 10))


def test_the_engines_reordering_keeps_every_dependency_of_every_program():
    """The engine holds weight gradients back (one launch for several) and launches two independent GEMMs together:
    grlhip_plan_schedule_check replays the launch schedule of a program without a GPU and checks every read-after-write,
    write-after-write and write-after-read pair of ops against the program order.  Checked on the nets of the test suite and
    on a sweep of other shapes (1-4 levels, widths that are and are not multiples of 128, with and without condition),
    for all four programs; the backward program must in fact be reordered (fewer launches than ops)."""
    import itertools
    import random
    from generativerl_preview_amd.engine import NetPlan
    lib = L.lib()
    rng = random.Random(7)
    shapes = [(6, 17, [512, 256, 128]), (2, 0, [512, 256, 128]), (64, 0, [1024]), (3, 11, [64, 32, 16]), (6, 44, [256, 128])]
    for _ in range(40):
        levels = rng.randint(1, 4)
        last = rng.choice([8, 12, 16, 24, 32, 40, 64, 72, 128])          # the U-Net wiring halves the width per level
        hidden = [last << (levels - 1 - i) for i in range(levels)]
        shapes.append((rng.randint(1, 40), rng.choice([0, 0, 5, 17, 33]), hidden))
    reordered = 0
    for x_dim, cond, hidden in shapes:
        plan = NetPlan(L.NET_RESNET, x_dim, x_dim, cond, 32, 32 if cond else 0, 128, hidden)
        for split, prog in itertools.product((3, 2), range(4)):
            plan.set_backward_split(split)
            assert lib.grlhip_plan_schedule_check(plan.handle, prog) == 0, (x_dim, cond, hidden, split, prog)
        n_ops, n_launch = ctypes.c_int32(), ctypes.c_int32()
        L.check(lib.grlhip_plan_program_info(plan.handle, 3, ctypes.byref(n_ops), ctypes.byref(n_launch)))
        reordered += int(n_launch.value < n_ops.value)
    assert reordered == len(shapes)
    for kind, args in ((L.NET_CONCAT_MLP, (6, 1, 17, 32, 0, 0, [256, 256])), (L.NET_QNET, (6, 1, 17, 0, 0, 0, [256, 256])),
                       (L.NET_CONCAT_MLP, (3, 1, 11, 32, 0, 0, [64, 48, 32]))):
        plan = NetPlan(kind, *args)
        for prog in range(4):
            assert lib.grlhip_plan_schedule_check(plan.handle, prog) == 0, (kind, args, prog)
    assert lib.grlhip_plan_schedule_check(None, 0) < 0
