"""DPM-Solver (grl/numerical_methods/numerical_solvers/dpm_solver.py:13-905), the solver the QGPO benchmark
configurations ship with (order 2, 17 function evaluations).

The per-step algebra is a handful of scalar schedule values (half-logSNR, log-scale, std) computed on the CPU in
fp32 exactly as the reference computes them on tensors.  Orders 1-3 of both solver types become a stage table that
ONE ``grlhip_sample_stages`` call runs on the device (``_integrate_fused``; the last stage of order 3 "taylor" uses
the second saved model value, ``W2``).  With ``with_grad=True`` the same table runs in fp32 with every stage state kept and
the backward is ONE ``grlhip_sample_stages_backward`` call; a model function without ``coefficients`` is stepped from the host with
autograd on, each model evaluation being the differentiable HIP forward of the C ABI.
Supported: ``method="singlestep"`` with orders 1-3, ``type`` "dpm_solver" / "dpm_solver++", ``solver_type``
"dpm_solver" / "taylor", ``skip_type`` "time_uniform" / "time_quadratic".  The option values on which the
reference itself fails raise here too (recorded in tests/golden/dpm.npz "raises"): ``skip_type="logSNR"``
(TypeError), ``denoise=True`` (TypeError), ``method="singlestep_fixed"``; "multistep" and "adaptive" are not
implemented."""
from typing import List

import torch


class DPMSolver:
    def __init__(self, order: int, device: str, atol: float = 1e-5, rtol: float = 1e-5, steps: int = None,
                 type: str = "dpm_solver", method: str = "singlestep", solver_type: str = "dpm_solver",
                 skip_type: str = "time_uniform", denoise: bool = False):
        self.type = type
        assert self.type in ["dpm_solver", "dpm_solver++"]
        self.use_dpm_solver_plus_plus = self.type == "dpm_solver++"
        self.atol, self.rtol, self.steps, self.order = atol, rtol, steps, order
        assert self.order in [1, 2, 3]
        self.method = method
        assert self.method in ["singlestep", "multistep", "singlestep_fixed", "adaptive"]
        self.solver_type = solver_type
        assert self.solver_type in ["dpm_solver", "taylor"]
        self.default_high_order_solvers = self.solver_type == "dpm_solver"
        self.skip_type = skip_type
        assert self.skip_type in ["logSNR", "time_uniform", "time_quadratic"]
        self.denoise = denoise
        self.device = device
        self.nfe = 0
        self.fused = True  # orders <= 2 on an affine model function: all stages in one grlhip_sample_stages call

    # ---- dpm_solver.py:127-161
    def get_time_steps(self, t_T, t_0, N):
        if self.skip_type == "logSNR":
            raise TypeError("skip_type='logSNR': the reference fails here (torch.exp of a Python float, "
                            "dpm_solver.py:139-143); use 'time_uniform' or 'time_quadratic'")
        if self.skip_type == "time_uniform":
            return torch.linspace(t_T, t_0, N + 1)
        return torch.linspace(t_T ** 0.5, t_0 ** 0.5, N + 1).pow(2)

    # ---- dpm_solver.py:163-230
    @staticmethod
    def get_orders_for_singlestep_solver(steps: int, order: int) -> List[int]:
        if order == 3:
            K = steps // 3 + 1
            if steps % 3 == 0:
                return [3] * (K - 2) + [2, 1]
            if steps % 3 == 1:
                return [3] * (K - 1) + [1]
            return [3] * (K - 1) + [2]
        if order == 2:
            if steps % 2 == 0:
                return [2] * (steps // 2 - 1) + [1] * 2
            return [2] * (steps // 2) + [1]
        if order == 1:
            return [1] * steps
        raise ValueError("'order' must be '1' or '2' or '3'.")

    def integrate(self, diffusion_process, noise_function, data_prediction_function, x, condition=None,
                  steps: int = None, save_intermediate: bool = False, with_grad: bool = False):
        """dpm_solver.py:83-905.  ``x``: (B, ...) device tensor; returns the final state, or the stacked states
        (len(orders) + 1, B, ...) with ``save_intermediate=True``.  ``with_grad`` (the reference simply does not wrap
        the call in ``no_grad``, diffusion_model.py:261-279): the stage table with a grad_fn whose backward is one
        grlhip_sample_stages_backward call; for a model function without ``coefficients`` the stages run host-driven
        with autograd on (every model evaluation the differentiable HIP forward, the stage algebra torch)."""
        steps = steps if steps is not None else self.steps if self.steps is not None else 20
        if self.method != "singlestep":
            raise NotImplementedError("DPMSolver method '%s' (the reference's 'singlestep_fixed' fails on an unbound "
                                      "variable; 'multistep' / 'adaptive' are not part of this build)" % self.method)
        if self.denoise:
            raise TypeError("denoise=True: the reference fails here (denoise_fn is called without its condition "
                            "argument, dpm_solver.py:232-248,897)")
        plus, high = self.use_dpm_solver_plus_plus, self.default_high_order_solvers
        dp, B, dev = diffusion_process, x.shape[0], x.device

        def model_fn(xx, t):  # t: 0-dim CPU tensor
            self.nfe += 1
            tt = t.to(dev, torch.float32).expand(B).contiguous()
            return data_prediction_function(tt, xx, condition) if plus else noise_function(tt, xx, condition)

        lam, la, sig = dp.HalfLogSNR, dp.log_scale, dp.std
        f = lambda v: float(v)  # noqa: E731 -- schedule scalars stay fp32 tensors until they scale a state

        def first(x, s, t):
            h = lam(t) - lam(s)
            m = model_fn(x, s)
            if plus:
                return f(sig(t) / sig(s)) * x - f(torch.exp(la(t)) * torch.expm1(-h)) * m
            return f(torch.exp(la(t) - la(s))) * x - f(sig(t) * torch.expm1(h)) * m

        def second(x, s, t, r1):
            r1 = torch.tensor(0.5) if r1 is None else r1
            h = lam(t) - lam(s)
            s1 = dp.InverseHalfLogSNR(lam(s) + r1 * h).reshape(-1)[0]
            m = model_fn(x, s)
            if plus:
                p11, p1 = torch.expm1(-r1 * h), torch.expm1(-h)
                x1 = f(sig(s1) / sig(s)) * x - f(torch.exp(la(s1)) * p11) * m
                m1 = model_fn(x1, s1)
                at = torch.exp(la(t))
                xt = f(sig(t) / sig(s)) * x - f(at * p1) * m
                if high:
                    return xt - f((0.5 / r1) * at * p1) * (m1 - m)
                return xt + f((1.0 / r1) * at * ((torch.exp(-h) - 1.0) / h + 1.0)) * (m1 - m)
            p11, p1 = torch.expm1(r1 * h), torch.expm1(h)
            x1 = f(torch.exp(la(s1) - la(s))) * x - f(sig(s1) * p11) * m
            m1 = model_fn(x1, s1)
            xt = f(torch.exp(la(t) - la(s))) * x - f(sig(t) * p1) * m
            if high:
                return xt - f((0.5 / r1) * sig(t) * p1) * (m1 - m)
            return xt - f((1.0 / r1) * sig(t) * ((torch.exp(h) - 1.0) / h - 1.0)) * (m1 - m)

        def third(x, s, t, r1, r2):
            r1 = torch.tensor(1.0 / 3.0) if r1 is None else r1
            r2 = torch.tensor(2.0 / 3.0) if r2 is None else r2
            h = lam(t) - lam(s)
            s1 = dp.InverseHalfLogSNR(lam(s) + r1 * h).reshape(-1)[0]
            s2 = dp.InverseHalfLogSNR(lam(s) + r2 * h).reshape(-1)[0]
            m = model_fn(x, s)
            if plus:
                p11, p12, p1 = torch.expm1(-r1 * h), torch.expm1(-r2 * h), torch.expm1(-h)
                p22 = torch.expm1(-r2 * h) / (r2 * h) + 1.0
                p2 = p1 / h + 1.0
                p3 = p2 / h - 0.5
                x1 = f(sig(s1) / sig(s)) * x - f(torch.exp(la(s1)) * p11) * m
                m1 = model_fn(x1, s1)
                a2, at = torch.exp(la(s2)), torch.exp(la(t))
                x2 = f(sig(s2) / sig(s)) * x - f(a2 * p12) * m + f(r2 / r1 * a2 * p22) * (m1 - m)
                m2 = model_fn(x2, s2)
                xt = f(sig(t) / sig(s)) * x - f(at * p1) * m
                if high:
                    return xt + f((1.0 / r2) * at * p2) * (m2 - m)
                D10, D11 = f(1.0 / r1) * (m1 - m), f(1.0 / r2) * (m2 - m)
                D1, D2 = (f(r2) * D10 - f(r1) * D11) / f(r2 - r1), 2.0 * (D11 - D10) / f(r2 - r1)
                return xt + f(at * p2) * D1 - f(at * p3) * D2
            p11, p12, p1 = torch.expm1(r1 * h), torch.expm1(r2 * h), torch.expm1(h)
            p22 = torch.expm1(r2 * h) / (r2 * h) - 1.0
            p2 = p1 / h - 1.0
            p3 = p2 / h - 0.5
            x1 = f(torch.exp(la(s1) - la(s))) * x - f(sig(s1) * p11) * m
            m1 = model_fn(x1, s1)
            x2 = f(torch.exp(la(s2) - la(s))) * x - f(sig(s2) * p12) * m - f(r2 / r1 * sig(s2) * p22) * (m1 - m)
            m2 = model_fn(x2, s2)
            xt = f(torch.exp(la(t) - la(s))) * x - f(sig(t) * p1) * m
            if high:
                return xt - f((1.0 / r2) * sig(t) * p2) * (m2 - m)
            D10, D11 = f(1.0 / r1) * (m1 - m), f(1.0 / r2) * (m2 - m)
            D1, D2 = (f(r2) * D10 - f(r1) * D11) / f(r2 - r1), 2.0 * (D11 - D10) / f(r2 - r1)
            return xt - f(sig(t) * p2) * D1 - f(sig(t) * p3) * D2

        t_0, t_T = 0.00001, float(dp.t_max)
        orders = self.get_orders_for_singlestep_solver(steps=steps, order=self.order)
        ts = self.get_time_steps(t_T=t_T, t_0=t_0, N=steps)
        fn = data_prediction_function if plus else noise_function
        if self.fused and hasattr(fn, "coefficients") and x.is_cuda and x.dim() == 2:
            if with_grad:
                return self._integrate_fused_with_grad(dp, fn, x, condition, orders, ts, save_intermediate)
            return self._integrate_fused(dp, fn, x, condition, orders, ts, save_intermediate)
        x_list = [x.clone()] if save_intermediate else None
        with torch.enable_grad() if with_grad else torch.no_grad():
            i = 0
            for o in orders:
                s, t = ts[i], ts[i + o]
                h = lam(ts[i + o]) - lam(ts[i])
                r1 = None if o <= 1 else (lam(ts[i + 1]) - lam(ts[i])) / h
                r2 = None if o <= 2 else (lam(ts[i + 2]) - lam(ts[i])) / h
                x = first(x, s, t) if o == 1 else second(x, s, t, r1) if o == 2 else third(x, s, t, r1, r2)
                if save_intermediate:
                    x_list.append(x.clone())
                i += o
        return torch.stack(x_list, dim=0) if save_intermediate else x

    def _integrate_fused(self, dp, fn, x, condition, orders, ts, save_intermediate):
        from ... import _lib as L
        table = self.stage_table(dp, fn, orders, ts, save_intermediate)
        assert table.shape[1] == L.COEF_STRIDE
        self.nfe += table.shape[0]
        m = fn.model
        dtype = m.compute_dtype
        g_plan = g_packed = None
        if fn.guidance is not None:
            g_plan, g_packed = fn.guidance.plan, fn.guidance.packed(dtype)
        final, traj = m.plan.sample_stages(m.packed(dtype), x, condition, table, guidance=g_plan, guidance_packed=g_packed,
                                           trajectory_slots=len(orders) + 1 if save_intermediate else 0, dtype=dtype)
        return traj if save_intermediate else final

    def _integrate_fused_with_grad(self, dp, fn, x, condition, orders, ts, save_intermediate):
        """``with_grad=True``: the same stage table in fp32, every stage's state kept, and a grad_fn whose backward is
        ONE grlhip_sample_stages_backward call (gradients reach the model's parameters and ``x``)."""
        from ...generative_models.fused import differentiable_stages
        table = self.stage_table(dp, fn, orders, ts, True)
        self.nfe += table.shape[0]
        states = differentiable_stages(fn, condition, table, x)
        ends = [0] + [i + 1 for i in range(table.shape[0]) if table[i, 9] >= 0]   # stages that end a solver step
        return states[ends] if save_intermediate else states[-1]

    def stage_table(self, dp, fn, orders, ts, save_intermediate=True):
        """Orders 1-3, both solver types: every stage is
        `m = cn net + cx x + cg dE/dx; x <- P x + Q xs + R m + W ms + W2 ms2` (include/grlhip.h, grlhip_sample_stages;
        ms2, the second saved model value, is only needed by the last stage of the "taylor" form of order 3); the same
        scalars as the host-driven path, tabulated once."""
        plus, high = self.use_dpm_solver_plus_plus, self.default_high_order_solvers
        lam, la, sig = dp.HalfLogSNR, dp.log_scale, dp.std
        rows = []

        def stage(tau, P, Q, R, W, save, emit, W2=0.0):
            cn, cx, cg = fn.coefficients(tau)
            rows.append([float(tau), float(P), float(Q), float(R), float(W), cn, cx, cg, float(save), float(emit), float(W2), 0.0])

        i = 0
        for k, o in enumerate(orders):
            s, t = ts[i], ts[i + o]
            h = lam(t) - lam(s)
            emit = k + 1 if save_intermediate else -1
            keep = (sig(t) / sig(s)) if plus else torch.exp(la(t) - la(s))                     # coefficient of x_s
            lead = (torch.exp(la(t)) * torch.expm1(-h)) if plus else (sig(t) * torch.expm1(h))  # coefficient of m_s
            if o == 1:
                stage(s, keep, 0.0, -lead, 0.0, 0, emit)
            elif o == 3:
                r1, r2 = (lam(ts[i + 1]) - lam(s)) / h, (lam(ts[i + 2]) - lam(s)) / h
                s1 = dp.InverseHalfLogSNR(lam(s) + r1 * h).reshape(-1)[0]
                s2 = dp.InverseHalfLogSNR(lam(s) + r2 * h).reshape(-1)[0]
                if plus:
                    a2, at = torch.exp(la(s2)), torch.exp(la(t))
                    p12, p1 = torch.expm1(-r2 * h), torch.expm1(-h)
                    c2 = r2 / r1 * a2 * (p12 / (r2 * h) + 1.0)            # x2 = .. + c2 (m1 - m)
                    c3 = (1.0 / r2) * at * (p1 / h + 1.0)                  # x_t = .. + c3 (m2 - m)
                    stage(s, sig(s1) / sig(s), 0.0, -(torch.exp(la(s1)) * torch.expm1(-r1 * h)), 0.0, 1, -1)
                    stage(s1, 0.0, sig(s2) / sig(s), c2, -(a2 * p12) - c2, 0 if high else 2, -1)
                    if high:
                        stage(s2, 0.0, keep, c3, -lead - c3, 0, emit)
                    else:
                        # x_t = keep x - lead m + at p2 D1 - at p3 D2 with D1, D2 linear in (m1 - m), (m2 - m)
                        p2 = p1 / h + 1.0
                        p3 = p2 / h - 0.5
                        k1 = at * p2 * (r2 / r1) / (r2 - r1) + at * p3 * 2.0 / (r1 * (r2 - r1))     # of (m1 - m)
                        k2 = -at * p2 * (r1 / r2) / (r2 - r1) - at * p3 * 2.0 / (r2 * (r2 - r1))    # of (m2 - m)
                        stage(s2, 0.0, keep, k2, -lead - k1 - k2, 0, emit, W2=k1)
                else:
                    p12, p1 = torch.expm1(r2 * h), torch.expm1(h)
                    c2 = r2 / r1 * sig(s2) * (p12 / (r2 * h) - 1.0)        # x2 = .. - c2 (m1 - m)
                    c3 = (1.0 / r2) * sig(t) * (p1 / h - 1.0)              # x_t = .. - c3 (m2 - m)
                    stage(s, torch.exp(la(s1) - la(s)), 0.0, -(sig(s1) * torch.expm1(r1 * h)), 0.0, 1, -1)
                    stage(s1, 0.0, torch.exp(la(s2) - la(s)), -c2, -(sig(s2) * p12) + c2, 0 if high else 2, -1)
                    if high:
                        stage(s2, 0.0, keep, -c3, -lead + c3, 0, emit)
                    else:
                        # x_t = keep x - lead m - sig(t) p2 D1 - sig(t) p3 D2
                        st_ = sig(t)
                        p2 = p1 / h - 1.0
                        p3 = p2 / h - 0.5
                        k1 = -st_ * p2 * (r2 / r1) / (r2 - r1) + st_ * p3 * 2.0 / (r1 * (r2 - r1))   # of (m1 - m)
                        k2 = st_ * p2 * (r1 / r2) / (r2 - r1) - st_ * p3 * 2.0 / (r2 * (r2 - r1))    # of (m2 - m)
                        stage(s2, 0.0, keep, k2, -lead - k1 - k2, 0, emit, W2=k1)
            else:
                r1 = (lam(ts[i + 1]) - lam(s)) / h
                s1 = dp.InverseHalfLogSNR(lam(s) + r1 * h).reshape(-1)[0]
                if plus:
                    stage(s, sig(s1) / sig(s), 0.0, -(torch.exp(la(s1)) * torch.expm1(-r1 * h)), 0.0, 1, -1)
                    if high:
                        c = (0.5 / r1) * torch.exp(la(t)) * torch.expm1(-h)      # x_t = .. - c (m1 - m)
                        stage(s1, 0.0, keep, -c, -lead + c, 0, emit)
                    else:
                        c = (1.0 / r1) * torch.exp(la(t)) * ((torch.exp(-h) - 1.0) / h + 1.0)  # x_t = .. + c (m1 - m)
                        stage(s1, 0.0, keep, c, -lead - c, 0, emit)
                else:
                    stage(s, torch.exp(la(s1) - la(s)), 0.0, -(sig(s1) * torch.expm1(r1 * h)), 0.0, 1, -1)
                    if high:
                        c = (0.5 / r1) * sig(t) * torch.expm1(h)
                    else:
                        c = (1.0 / r1) * sig(t) * ((torch.exp(h) - 1.0) / h - 1.0)
                    stage(s1, 0.0, keep, -c, -lead + c, 0, emit)
            i += o
        return torch.tensor(rows, dtype=torch.float32)
