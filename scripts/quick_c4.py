"""C4 on the bf16 engine: QGPO net + fused energy guidance, guided ODE and guided Euler-Maruyama."""
import json, sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests/golden')
import golden_lib as gl
import generativerl_preview_amd as G
from generativerl_preview_amd import _lib as L
from generativerl_preview_amd.numerical_methods.numerical_solvers import SDESolver
DEV = "cuda:0"


def timed(fn, n=2):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n


ts = torch.linspace(0, 1, 101)
q = G.EnergyConditionalDiffusionModel(dict(gl.diffusion_cfg(6, gl.VP, "noise_function", [512, 256, 128], 17, energy=True), device=DEV)).to(DEV)
q.model.compute_dtype = L.BF16
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
cond = torch.randn(R, 17, device=DEV); x0 = torch.randn(R, 6, device=DEV)
t = timed(lambda: q.sample(t_span=ts, x_0=x0, condition=cond, guidance_scale=1.0))
print(json.dumps(dict(config="C4 QGPO guided ODE 100 steps, %d rows" % R, engine="bf16", seconds=t,
                      action_steps_per_s=R * 100 / t, tflops=R * 100 * 3547712 / t / 1e12)), flush=True)
n_inner = len(SDESolver(dt=0.01)._schedule(ts)[0])
dW = torch.randn(n_inner, R, 6, device=DEV) * 0.1
t = timed(lambda: q.sample(t_span=ts, x_0=x0, condition=cond, guidance_scale=1.0, noise_increments=dW,
                           solver_config=dict(type="SDESolver", args=dict(dt=0.01))))
print(json.dumps(dict(config="C4 QGPO guided Euler-Maruyama %d steps, %d rows" % (n_inner, R), engine="bf16", seconds=t,
                      action_steps_per_s=R * n_inner / t, tflops=R * n_inner * 3547712 / t / 1e12)), flush=True)

# C5 on the batch engine with bf16 MFMAs (no fused kernel for the [1024] net)
g5 = G.DiffusionModel(dict(gl.diffusion_cfg(64, gl.GVP, "velocity_function", [1024]), device=DEV)).to(DEV)
g5.model.compute_dtype = L.BF16
R5 = 262144
t = timed(lambda: g5.sample(t_span=ts, batch_size=R5), 1)
print(json.dumps(dict(config="C5 GVP [1024] D=64, 100-step ODE, batch 262144", engine="bf16 (batch engine)", seconds=t,
                      action_steps_per_s=R5 * 100 / t, tflops=R5 * 100 * 6950912 / t / 1e12)), flush=True)

# QGPO's shipped solver: DPM-Solver-2 singlestep, 17 function evaluations, energy-guided (fp32 batch engine per NFE)
q.model.compute_dtype = L.F32
Rd = 1 << 20
condd = torch.randn(Rd, 17, device=DEV); x0d = torch.randn(Rd, 6, device=DEV)
sc = dict(type="DPMSolver", args=dict(order=2, device=DEV, steps=17))
t = timed(lambda: q.sample(x_0=x0d, condition=condd, guidance_scale=1.0, solver_config=sc), 1)
print(json.dumps(dict(config="QGPO guided DPM-Solver-2, 17 NFE, %d rows" % Rd, engine="f32 batch engine (grlhip_sample_stages)", seconds=t,
                      samples_per_s=Rd / t, action_steps_per_s=Rd * 17 / t, tflops=Rd * 17 * 3547712 / t / 1e12)), flush=True)

q.model.compute_dtype = L.BF16
Rb = 1 << 22
condb = torch.randn(Rb, 17, device=DEV); x0b = torch.randn(Rb, 6, device=DEV)
for typ in ("dpm_solver", "dpm_solver++"):
    scb = dict(type="DPMSolver", args=dict(order=2, device=DEV, steps=17, type=typ))
    t = timed(lambda: q.sample(x_0=x0b, condition=condb, guidance_scale=1.0, solver_config=scb), 2)
    print(json.dumps(dict(config="QGPO guided DPM-Solver-2 (%s), 17 NFE, %d rows" % (typ, Rb), engine="bf16 persistent integrator (stage mode)",
                          seconds=t, samples_per_s=Rb / t, action_steps_per_s=Rb * 17 / t, tflops=Rb * 17 * 3547712 / t / 1e12)), flush=True)
