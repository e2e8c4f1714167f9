"""grlhip_sample_ode_static (include/grlhip.h): the allocation-free variant of grlhip_sample_ode whose scratch is a
device array inside the library.  Called through the C ABI (ctypes) and compared BIT FOR BIT with grlhip_sample_ode
on the same inputs -- same kernels, same arithmetic, only the scratch differs -- on two calls in a row, the second
with a smaller input (fewer rows, fewer steps), so stale scratch contents of the first call would show."""
import ctypes

import pytest
import torch

import util_models as um
from generativerl_preview_amd import _lib as L
from generativerl_preview_amd.engine import _ptr, _stream
from generativerl_preview_amd.generative_models.diffusion_process import GuidedScore

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _table(m, n_steps, guided):
    """Euler coefficient table of the reverse ODE on linspace(0, 1, n_steps + 1), finals only."""
    t_span = torch.linspace(0.0, 1.0, n_steps + 1)
    if guided:
        fn = GuidedScore(m.model, "noise_function", m.energy_guidance.model, 1.0)
        drift = m.diffusion_process.reverse_ode(fn, "score_function").drift
    else:
        drift = m.diffusion_process.reverse_ode(m.model, "noise_function").drift
    dts = t_span[1:] - t_span[:-1]
    none = torch.full((n_steps,), -1.0)
    return drift.table(t_span[:-1], dts, none, torch.ones(n_steps)).to(DEV)


def _both(m, g, rows, n_steps, seed):
    lib = L.lib()
    gen = torch.Generator().manual_seed(seed)
    x0 = torch.randn(rows, 6, generator=gen).to(DEV)
    cond = torch.randn(rows, 17, generator=gen).to(DEV)
    coef = _table(m, n_steps, g is not None)
    plan, packed = m.model.plan, m.model.packed(L.BF16)
    g_h = g.plan.handle if g is not None else None
    g_packed = g.packed(L.BF16) if g is not None else None
    need = lib.grlhip_workspace_bytes(plan.handle, g_h, rows, L.BF16)
    assert need <= lib.grlhip_static_scratch_bytes()
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    want = torch.full((rows, 6), float("nan"), device=DEV)
    got = torch.full((rows, 6), float("nan"), device=DEV)
    with torch.cuda.device(DEV):
        L.check(lib.grlhip_sample_ode(plan.handle, _ptr(packed), L.BF16, g_h, _ptr(g_packed), _ptr(x0), _ptr(cond),
                                      _ptr(coef), n_steps, rows, _ptr(want), None, _ptr(ws), ws.numel(), _stream(DEV)))
        torch.cuda.synchronize()
        def static_call():
            L.check(lib.grlhip_sample_ode_static(plan.handle, _ptr(packed), g_h, _ptr(g_packed), _ptr(x0), _ptr(cond),
                                                 _ptr(coef), n_steps, rows, _ptr(got), None, _stream(DEV)))
            torch.cuda.synchronize()
        static_call()    # the first call on a device resolves the scratch symbol (may load the code object)
        got.fill_(float("nan"))
        torch.cuda.synchronize()
        # torch's counter only sees torch's caching allocator; the driver's free-byte count (hipMemGetInfo) also
        # sees a hipMalloc / hipFree made inside the library
        before, free_before = torch.cuda.memory_allocated(DEV), torch.cuda.mem_get_info(DEV)[0]
        static_call()
        assert torch.cuda.memory_allocated(DEV) == before
        assert torch.cuda.mem_get_info(DEV)[0] == free_before
    assert bool(torch.isfinite(want).all())
    assert torch.equal(got, want), float((got - want).abs().max())


def test_static_scratch_entry_point_matches_sample_ode_twice():
    kind, cfg, m = um.build("qgpo_noise", DEV)
    _both(m, None, 1000, 20, seed=11)
    _both(m, None, 333, 7, seed=12)      # smaller input on the same scratch


def test_static_scratch_entry_point_guided_twice():
    kind, cfg, m = um.build("qgpo_noise", DEV)
    g = m.energy_guidance.model
    _both(m, g, 700, 12, seed=13)
    _both(m, g, 129, 5, seed=14)


def test_static_scratch_rejects_what_it_cannot_hold():
    lib = L.lib()
    kind, cfg, m = um.build("gvp_wide", DEV)   # no fused bf16 integrator for this shape
    plan = m.model.plan
    x0 = torch.zeros(4, 64, device=DEV)
    out = torch.zeros(4, 64, device=DEV)
    coef = torch.zeros(1, L.COEF_STRIDE, device=DEV)
    rc = lib.grlhip_sample_ode_static(plan.handle, _ptr(m.model.packed(L.F32)), None, None, _ptr(x0), None, _ptr(coef),
                                      1, 4, _ptr(out), None, _stream(DEV))
    assert rc == -2, (rc, lib.grlhip_last_error())


@pytest.mark.parametrize("guided", [False, True])
def test_static_scratch_entry_point_under_hip_graph_capture(guided):
    """The caller the entry point exists for: grlhip_sample_ode_static captured into a HIP graph (after one call outside
    the capture, which resolves the scratch symbol) and replayed on new inputs written into the same buffers; every
    replay equals grlhip_sample_ode on those inputs bit for bit."""
    lib = L.lib()
    kind, cfg, m = um.build("qgpo_noise", DEV)
    g = m.energy_guidance.model if guided else None
    rows, n_steps = 515, 9
    coef = _table(m, n_steps, guided)
    plan, packed = m.model.plan, m.model.packed(L.BF16)
    g_h = g.plan.handle if guided else None
    g_packed = g.packed(L.BF16) if guided else None
    x0 = torch.zeros(rows, 6, device=DEV)
    cond = torch.zeros(rows, 17, device=DEV)
    got = torch.zeros(rows, 6, device=DEV)
    want = torch.zeros(rows, 6, device=DEV)
    need = lib.grlhip_workspace_bytes(plan.handle, g_h, rows, L.BF16)
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)

    def static_call():
        L.check(lib.grlhip_sample_ode_static(plan.handle, _ptr(packed), g_h, _ptr(g_packed), _ptr(x0), _ptr(cond),
                                             _ptr(coef), n_steps, rows, _ptr(got), None, _stream(DEV)))
    with torch.cuda.device(DEV):
        static_call()                     # outside any capture: resolves the scratch symbol on this device
        torch.cuda.synchronize()
        side = torch.cuda.Stream(device=DEV)
        side.wait_stream(torch.cuda.current_stream(DEV))
        with torch.cuda.stream(side):
            static_call()
        torch.cuda.current_stream(DEV).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_call()
        gen = torch.Generator().manual_seed(21)
        for _ in range(3):
            x0.copy_(torch.randn(rows, 6, generator=gen))
            cond.copy_(torch.randn(rows, 17, generator=gen))
            got.fill_(float("nan"))
            graph.replay()
            L.check(lib.grlhip_sample_ode(plan.handle, _ptr(packed), L.BF16, g_h, _ptr(g_packed), _ptr(x0), _ptr(cond),
                                          _ptr(coef), n_steps, rows, _ptr(want), None, _ptr(ws), ws.numel(), _stream(DEV)))
            torch.cuda.synchronize()
            assert bool(torch.isfinite(want).all())
            assert torch.equal(got, want), float((got - want).abs().max())
