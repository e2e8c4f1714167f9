"""ODESolver (grl/numerical_methods/numerical_solvers/ode_solver.py:17-185).

Same constructor schema (``ode_solver="euler", dt, atol, rtol, library, **kwargs``) and
``integrate(drift, x0, t_span)`` -> (T, B, D) trajectory.  The reference delegates the stepping to
torchdyn / torchdiffeq; for ``euler`` both are fixed-grid explicit Euler on exactly the caller's
grid (atol/rtol/dt unused), which is what the fused HIP integrator implements in one launch when
``drift`` is a ``FusedDrift``.  A plain Python callable is stepped on the host, one call per step."""
import torch

from ...generative_models.fused import FusedDrift


class ODESolver:
    def __init__(self, ode_solver="euler", dt=0.01, atol=1e-5, rtol=1e-5, library="torchdyn", **kwargs):
        self.ode_solver, self.dt, self.atol, self.rtol = ode_solver, dt, atol, rtol
        self.nfe = 0
        self.kwargs = kwargs
        self.library = library

    def integrate(self, drift, x0, t_span, return_trajectory=True, with_grad=False, **kwargs):
        self.nfe = 0
        if self.library not in ("torchdiffeq", "torchdiffeq_adjoint", "torchdyn", "torchdyn_NeuralODE"):
            raise ValueError(f"library {self.library} is not supported")
        if self.ode_solver != "euler":
            raise NotImplementedError("only the explicit Euler scheme is implemented (ode_solver='euler')")
        n_steps = len(t_span) - 1
        if isinstance(drift, FusedDrift):
            ts = t_span.detach().to("cpu", torch.float32)
            table = drift.table(ts[:-1], ts[1:] - ts[:-1], torch.arange(1, n_steps + 1), torch.ones(n_steps))
            if with_grad:
                traj = drift.integrate_with_grad(x0, table)
                self.nfe = n_steps
                return traj if return_trajectory else traj[-1]
            final, traj = drift.integrate(x0, table, None, trajectory_slots=(n_steps + 1) if return_trajectory else 0)
            self.nfe = n_steps
            return traj if return_trajectory else final
        x = x0
        out = [x0]
        for k in range(n_steps):
            self.nfe += 1
            x = x + (t_span[k + 1] - t_span[k]) * drift(t_span[k].repeat(x.shape[0]), x)
            out.append(x)
        return torch.stack(out) if return_trajectory else x
