"""SDESolver (grl/numerical_methods/numerical_solvers/sde_solver.py:52-144).

Same constructor schema.  The reference calls ``torchsde.sdeint(method="euler", dt=dt)``: the step
size is ``dt`` (not the ``t_span`` spacing): t advances ``min(t + dt, t_span[-1])`` from
``t_span[0]``, each step is diagonal-Ito Euler-Maruyama ``y += f dt + g dW``, and the outputs at
``t_span[k]`` are linearly interpolated between the two bracketing steps.  That schedule is
tabulated on the host (``_schedule``) and executed by the fused HIP integrator in one launch.

Reproducibility superset: ``noise_increments=`` (n_inner_steps, B, D), already scaled by sqrt(dt_k),
or ``bm=`` (callable ``bm(t_a, t_b) -> (B, D)`` like a torchsde BrownianInterval) pin the Brownian
path; without them the increments are drawn with ``torch.randn`` on the device."""
import torch

from ...generative_models.fused import FusedDrift


class SDESolver:
    def __init__(self, sde_solver="euler", sde_noise_type="diagonal", sde_type="ito", dt=0.001, atol=1e-5,
                 rtol=1e-5, library="torchsde", **kwargs):
        self.sde_solver, self.sde_noise_type, self.sde_type = sde_solver, sde_noise_type, sde_type
        self.dt, self.atol, self.rtol = dt, atol, rtol
        self.nfe_drift = 0
        self.nfe_diffusion = 0
        self.kwargs = kwargs
        self.library = library

    def _schedule(self, t_span):
        """torchsde's stepping, tabulated: inner steps (t0, dt) and, per requested output k >= 1, the
        step that brackets it and the interpolation weight w (out = y_prev + w (y_new - y_prev))."""
        ts = t_span.detach().to("cpu", torch.float32)
        dt = torch.as_tensor(self.dt, dtype=ts.dtype)
        t0s, dts, outs = [], [], []   # outs: (output slot, step index, w)
        curr, prev = ts[0], ts[0]
        for k in range(1, len(ts)):
            out_t = ts[k]
            while curr < out_t:
                nxt = torch.minimum(curr + dt, ts[-1])
                t0s.append(curr)
                dts.append(nxt - curr)
                prev, curr = curr, nxt
            if not t0s:
                raise ValueError("t_span must be increasing")
            w = 1.0 if (curr == out_t or prev == curr) else float((out_t - prev) / (curr - prev))
            outs.append((k, len(t0s) - 1, w))
        return torch.stack(t0s), torch.stack(dts), outs

    @staticmethod
    def _emit_plan(n_steps, outs):
        """The kernel emits at most two states per step; further outputs bracketed by the same step are
        linear in those two and are filled in afterwards (`extra`: (slot, slotA, slotB, coefficient))."""
        emit, lerp = torch.full((n_steps,), -1.0), torch.ones(n_steps)
        emit2, lerp2 = torch.full((n_steps,), -1.0), torch.ones(n_steps)
        by_step = {}
        for k, s, w in outs:
            by_step.setdefault(s, []).append((k, w))
        extra = []
        for s, lst in by_step.items():
            emit[s], lerp[s] = lst[0]
            if len(lst) > 1:
                emit2[s], lerp2[s] = lst[-1]
                (ka, wa), (kb, wb) = lst[0], lst[-1]
                for k, w in lst[1:-1]:
                    extra.append((k, ka, kb, 0.0 if wb == wa else (w - wa) / (wb - wa)))
        return emit, lerp, emit2, lerp2, extra

    def integrate(self, drift, diffusion, x0, t_span, logqp=False, adaptive=False, noise_increments=None,
                  return_trajectory=True, with_grad=False):
        if self.sde_solver != "euler" or self.sde_noise_type != "diagonal" or self.sde_type != "ito" or adaptive or logqp:
            raise NotImplementedError("only the fixed-step diagonal-Ito Euler-Maruyama scheme is implemented")
        if not isinstance(drift, FusedDrift):
            raise NotImplementedError("SDESolver integrates the reverse SDEs built by DiffusionProcess.reverse_sde")
        t0s, dts, outs = self._schedule(t_span)
        n = len(t0s)
        emit, lerp, emit2, lerp2, extra = self._emit_plan(n, outs)
        B = x0.shape[0]
        flat = x0.reshape(B, -1)
        if noise_increments is None:
            bm = self.kwargs.get("bm", None)
            if bm is not None:
                noise_increments = torch.stack([bm(t0s[k], t0s[k] + dts[k]).to(flat.device, torch.float32).reshape(B, -1)
                                                for k in range(n)])
            else:
                noise_increments = torch.randn(n, *flat.shape, device=flat.device) * dts.sqrt().to(flat.device)[:, None, None]
        if with_grad:
            # differentiable: keep every inner state (slot k + 1 after inner step k), walk the chain backwards in the
            # library (the increments are constants of the chain), interpolate the requested outputs in torch
            every = drift.table(t0s, dts, torch.arange(1, n + 1, dtype=torch.float32), torch.ones(n))
            inner = drift.integrate_with_grad(flat, every, noise_increments.reshape(n, B, -1))
            self.nfe_drift = self.nfe_diffusion = n
            out = [inner[0]] + [inner[s] + w * (inner[s + 1] - inner[s]) for _, s, w in outs]
            if not return_trajectory:
                return out[-1].reshape(x0.shape)
            return torch.stack(out, 0).reshape(t_span.shape[0], *x0.shape)
        table = drift.table(t0s, dts, emit, lerp, emit2, lerp2)
        final, traj = drift.integrate(flat, table, noise_increments.reshape(n, B, -1),
                                      trajectory_slots=len(t_span) if return_trajectory else 0)
        self.nfe_drift = self.nfe_diffusion = n
        if not return_trajectory:
            return final.reshape(x0.shape)
        for k, ka, kb, c in extra:
            traj[k] = traj[ka] + c * (traj[kb] - traj[ka])
        return traj.reshape(t_span.shape[0], *x0.shape)
