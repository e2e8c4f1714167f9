"""Throughput of the BASELINE configs other than the bench line (C1, C3, C4, C5), one JSON line each.
C1/C5/C4 run on the engines that support them today."""
import json, sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests/golden')
import golden_lib as gl
import generativerl_preview_amd as G
from generativerl_preview_amd import _lib as L
DEV = "cuda:0"


def timed(fn, n=2):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n


def report(name, **kw):
    print(json.dumps(dict(config=name, **kw)), flush=True)


ts = torch.linspace(0, 1, 101)
# C1: swiss-roll net, 999-step ODE, batch 500 (latency-bound) and a throughput-sized batch
m = G.DiffusionModel(dict(gl.diffusion_cfg(2, gl.VP, "noise_function", [512, 256, 128]), device=DEV)).to(DEV)
for dt_name, dt in (("f32", L.F32), ("bf16", L.BF16)):
    m.model.compute_dtype = dt
    t = timed(lambda: m.sample(t_span=torch.linspace(0, 1, 1000), batch_size=500))
    report("C1 swiss-roll 999-step ODE, batch 500", engine=dt_name, seconds=t, action_steps_per_s=500 * 999 / t)
m.model.compute_dtype = L.BF16
rows = 1 << 20
t = timed(lambda: m.sample(t_span=ts, batch_size=rows))
report("C1 net, 100-step ODE, 1M rows", engine="bf16", seconds=t, action_steps_per_s=rows * 100 / t,
       tflops=rows * 100 * 2846720 / t / 1e12)
opt = torch.optim.Adam(m.parameters(), lr=5e-3)
x = torch.randn(4096, 2, device=DEV)
def c1_train():
    opt.zero_grad(); loss = m.score_matching_loss(x); loss.backward()
    torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0); opt.step()
t = timed(c1_train, 5)
report("C1 training step (score_matching_loss default weighting + clip + Adam), batch 4096", engine="f32", seconds=t, rows_per_s=4096 / t)
# the same loop on flat buffers: FlatAdam (one launch) and the clip as one norm over the flat gradient buffer
from generativerl_preview_amd.optim import FlatAdam
m1 = G.DiffusionModel(dict(gl.diffusion_cfg(2, gl.VP, "noise_function", [512, 256, 128]), device=DEV)).to(DEV)
fopt = FlatAdam(m1.model, lr=5e-3)
def c1_train_flat():
    fopt.zero_grad(); loss = m1.score_matching_loss(x); loss.backward()
    fopt.fg.clip_grad_norm_(1.0); fopt.step()
t = timed(c1_train_flat, 5)
report("C1 training step, the same with FlatAdam + FlatGrads.clip_grad_norm_, batch 4096", engine="f32", seconds=t, rows_per_s=4096 / t)

# C3: ICFM training step, batch 65536
icfm = G.IndependentConditionalFlowModel(dict(gl.icfm_cfg(6, [512, 256, 128], 17), device=DEV)).to(DEV)
opt3 = torch.optim.Adam(icfm.parameters(), lr=1e-4)
B = 65536
x1 = torch.rand(B, 6, device=DEV) * 2 - 1; c = torch.randn(B, 17, device=DEV)
def c3():
    opt3.zero_grad(); loss = icfm.flow_matching_loss(torch.randn(B, 6, device=DEV), x1, c); loss.backward(); opt3.step()
t = timed(c3, 3)
report("C3 ICFM flow_matching_loss fwd+bwd+Adam, batch 65536", engine="f32", seconds=t, rows_per_s=B / t, tflops=B * 9759936 / t / 1e12)

# C4: QGPO energy-guided Euler-Maruyama, dt = 1/100, fixed increments (one GPU's shard at reduced rows)
q = G.EnergyConditionalDiffusionModel(dict(gl.diffusion_cfg(6, gl.VP, "noise_function", [512, 256, 128], 17, energy=True), device=DEV)).to(DEV)
R = 1 << 18
cond = torch.randn(R, 17, device=DEV); x0 = torch.randn(R, 6, device=DEV)
from generativerl_preview_amd.numerical_methods.numerical_solvers import SDESolver
n_inner = len(SDESolver(dt=0.01)._schedule(ts)[0])   # 100 or 101: fp32 accumulation of dt vs linspace
dW = torch.randn(n_inner, R, 6, device=DEV) * 0.1
t = timed(lambda: q.sample(t_span=ts, x_0=x0, condition=cond, guidance_scale=1.0, noise_increments=dW,
                           solver_config=dict(type="SDESolver", args=dict(dt=0.01))), 1)
report("C4 QGPO guided Euler-Maruyama %d steps, %d rows" % (n_inner, R), engine="f32", seconds=t,
       action_steps_per_s=R * n_inner / t, tflops=R * n_inner * 3547712 / t / 1e12)

# C5: GVP velocity net, D = 64, hidden [1024], batch 262144, 100-step ODE
g5 = G.DiffusionModel(dict(gl.diffusion_cfg(64, gl.GVP, "velocity_function", [1024]), device=DEV)).to(DEV)
R5 = 262144
t = timed(lambda: g5.sample(t_span=ts, batch_size=R5), 1)
report("C5 GVP [1024] D=64, 100-step ODE, batch 262144", engine="f32", seconds=t, action_steps_per_s=R5 * 100 / t,
       tflops=R5 * 100 * 6950912 / t / 1e12)
