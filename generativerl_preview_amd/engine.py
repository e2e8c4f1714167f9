"""Thin object layer over the C ABI: plans, packed weight images, workspaces."""
import ctypes

import torch

from . import _lib as L


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _f32c(t, name, shape=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise L.GrlHipError("%s must live on a HIP device (got %s): the hot path has no CPU fallback" % (name, t.device))
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    t = t.contiguous()
    if shape is not None:
        assert tuple(t.shape) == tuple(shape), "%s: expected shape %s, got %s" % (name, tuple(shape), tuple(t.shape))
    return t


class NetPlan:
    """One network description -> plan handle (grlhip_plan_create)."""

    def __init__(self, kind, x_dim, out_dim, cond_dim, t_dim, cond_hidden, t_cond_hidden, hidden, device=None):
        """`device`: the HIP device the plan's program tables are uploaded to (made current for the call)."""
        d = L.NetDesc()
        d.kind, d.x_dim, d.out_dim, d.cond_dim, d.t_dim = kind, x_dim, out_dim, cond_dim or 0, t_dim
        d.cond_hidden, d.t_cond_hidden, d.n_hidden = cond_hidden or 0, t_cond_hidden or 0, len(hidden)
        if len(hidden) > L.MAX_HIDDEN:
            raise ValueError("at most %d hidden sizes are supported" % L.MAX_HIDDEN)
        for i, h in enumerate(hidden):
            d.hidden[i] = int(h)
        self.desc = d
        self.kind, self.x_dim, self.out_dim, self.cond_dim = kind, x_dim, out_dim, cond_dim or 0
        h = ctypes.c_void_p()
        device = torch.device(device) if device is not None else None
        if device is not None and device.type == "cuda":
            with torch.cuda.device(device):
                L.check(L.lib().grlhip_plan_create(ctypes.byref(d), ctypes.byref(h)))
        else:
            L.check(L.lib().grlhip_plan_create(ctypes.byref(d), ctypes.byref(h)))
        self.device = device
        self.handle = h
        self._ws = {}
        self.fourier_grad = True     # the library's default (grlhip_plan_set_fourier_grad)
        self.backward_split = 3      # the library's default (grlhip_plan_set_backward_split)
        n = L.lib().grlhip_plan_num_params(h)
        self.params = []
        buf = ctypes.create_string_buffer(256)
        for i in range(n):
            r, c = ctypes.c_int32(), ctypes.c_int32()
            L.check(L.lib().grlhip_plan_param_info(h, i, buf, 256, ctypes.byref(r), ctypes.byref(c)))
            self.params.append((buf.value.decode(), (r.value, c.value) if c.value else (r.value,)))

    def set_fourier_grad(self, enable):
        """Whether the training entry points compute the gradient of the time encoder's frequencies."""
        if enable != self.fourier_grad:
            L.check(L.lib().grlhip_plan_set_fourier_grad(self.handle, int(enable)))
            self.fourier_grad = enable

    def set_backward_split(self, pieces):
        """bf16 pieces per fp32 value in the GEMMs of the backward programs: 3 (default, fp32 rounding level) or 2
        (three MFMAs per product instead of six: gradients carry ~1e-5 relative error per product)."""
        L.check(L.lib().grlhip_plan_set_backward_split(self.handle, int(pieces)))
        self.backward_split = int(pieces)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                L.lib().grlhip_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def fused_kernel(self, guidance=None):
        """Name of the persistent fused bf16 integrator that GRLHIP_BF16 sampling of this plan (energy-guided by the
        plan `guidance`) runs, or "" when it runs layer by layer on the batch engine (grlhip_bf16_fused_kernel)."""
        rc, text = self._bf16_kernel(guidance)
        return text if rc == 1 else ""

    def bf16_engine(self, guidance=None):
        """What GRLHIP_BF16 sampling of this plan runs: the fused integrator's kernel class, or the layer engine
        ("layer engine: k_w_gemm ..." for wide nets, "layer engine: k_bt_gemm_bf16 ..." otherwise)."""
        return self._bf16_kernel(guidance)[1]

    def _bf16_kernel(self, guidance):
        buf = ctypes.create_string_buffer(200)
        rc = L.lib().grlhip_bf16_fused_kernel(self.handle, guidance.handle if guidance is not None else None, buf, 200)
        if rc < 0:
            L.check(rc)
        return rc, buf.value.decode()

    def packed_bytes(self, dtype=L.F32):
        return L.lib().grlhip_packed_bytes(self.handle, dtype)

    def pack(self, named_tensors, dtype=L.F32):
        """named_tensors: mapping plan-relative name -> fp32 device tensor."""
        ts = []
        for name, shape in self.params:
            t = _f32c(named_tensors[name], name)
            if tuple(t.shape) != tuple(shape):
                raise ValueError("parameter %s: expected %s, got %s" % (name, shape, tuple(t.shape)))
            ts.append(t)
        dev = ts[0].device
        packed = torch.empty(max(self.packed_bytes(dtype), 256), dtype=torch.uint8, device=dev)
        arr = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        with torch.cuda.device(dev):
            L.check(L.lib().grlhip_pack_weights(self.handle, arr, len(ts), dtype, _ptr(packed), _stream(dev)))
        return packed

    def workspace(self, rows, device, guidance=None, dtype=L.F32):
        need = L.lib().grlhip_workspace_bytes(self.handle, guidance.handle if guidance else None, rows, dtype)
        key = (str(device), dtype)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(max(need, 256), dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws

    def train_workspace(self, rows, device):
        """Cached workspace of the training / backward entry points (gigabytes at 65 536 rows): grown on demand.
        One per (device, current stream): the dW partials that make the gradients bit-reproducible live in it, so two
        calls of one plan issued on different streams must not share a slab."""
        need = L.lib().grlhip_loss_workspace_bytes(self.handle, rows, L.F32)
        key = (str(device), "train", int(torch.cuda.current_stream(device).cuda_stream))
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            self._ws.pop(key, None)
            ws = None
            ws = torch.empty(max(need, 256), dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws

    def forward(self, packed, t, x, cond, dtype=L.F32):
        rows = x.shape[0]
        dev = x.device
        t, x, cond = _f32c(t, "t", (rows,)), _f32c(x, "x", (rows, self.x_dim)), _f32c(cond, "condition")
        out = torch.empty(rows, self.out_dim, dtype=torch.float32, device=dev)
        ws = self.workspace(rows, dev, None, dtype)
        with torch.cuda.device(dev):
            L.check(L.lib().grlhip_net_forward(self.handle, _ptr(packed), dtype, _ptr(t), _ptr(x), _ptr(cond),
                                               _ptr(out), rows, _ptr(ws), ws.numel(), _stream(dev)))
        return out

    def backward(self, packed, t, x, cond, grad_out, need_params=True, need_x=False):
        """grlhip_net_backward: (flat parameter gradient | None, d/dx | None) of sum(grad_out * net(t, x, cond));
        the forward is recomputed inside the call."""
        rows = x.shape[0]
        dev = x.device
        t, x, cond = _f32c(t, "t", (rows,)), _f32c(x, "x", (rows, self.x_dim)), _f32c(cond, "condition")
        grad_out = _f32c(grad_out, "grad_out", (rows, self.out_dim))
        flat = torch.zeros(L.lib().grlhip_grad_numel(self.handle), dtype=torch.float32, device=dev) if need_params else None
        gx = torch.empty(rows, self.x_dim, dtype=torch.float32, device=dev) if need_x else None
        ws = self.train_workspace(rows, dev)
        with torch.cuda.device(dev):
            L.check(L.lib().grlhip_net_backward(self.handle, _ptr(packed), L.F32, _ptr(t), _ptr(x), _ptr(cond),
                                                _ptr(grad_out), _ptr(flat), _ptr(gx), rows, _ptr(ws), ws.numel(),
                                                _stream(dev)))
        return flat, gx

    def sample_ode_backward(self, packed, traj, cond, coef, grad_traj):
        """grlhip_sample_ode_backward -> (flat parameter gradient, d loss / d x0)."""
        n1, rows, D = traj.shape
        dev = traj.device
        traj, grad_traj = _f32c(traj, "trajectory"), _f32c(grad_traj, "grad_traj", (n1, rows, D))
        cond, coef = _f32c(cond, "condition"), _f32c(coef.to(dev), "coef", (n1 - 1, L.COEF_STRIDE))
        flat = torch.zeros(L.lib().grlhip_grad_numel(self.handle), dtype=torch.float32, device=dev)
        gx0 = torch.empty(rows, D, dtype=torch.float32, device=dev)
        ws = self.train_workspace(rows, dev)
        with torch.cuda.device(dev):
            L.check(L.lib().grlhip_sample_ode_backward(self.handle, _ptr(packed), L.F32, _ptr(traj), _ptr(cond), _ptr(coef),
                                                       n1 - 1, rows, _ptr(grad_traj), _ptr(gx0), _ptr(flat), _ptr(ws),
                                                       ws.numel(), _stream(dev)))
        return flat, gx0

    def sample_stages_backward(self, packed, states, cond, table, grad_states):
        """grlhip_sample_stages_backward -> (flat parameter gradient, d loss / d x0).  states: (n_stages + 1, R, D),
        every stage's input state then the final one."""
        n1, rows, D = states.shape
        dev = states.device
        states, grad_states = _f32c(states, "stage states"), _f32c(grad_states, "grad_states", (n1, rows, D))
        cond, table = _f32c(cond, "condition"), _f32c(table.to(dev), "stage table", (n1 - 1, L.COEF_STRIDE))
        flat = torch.zeros(L.lib().grlhip_grad_numel(self.handle), dtype=torch.float32, device=dev)
        gx0 = torch.empty(rows, D, dtype=torch.float32, device=dev)
        ws = self.train_workspace(rows, dev)
        with torch.cuda.device(dev):
            L.check(L.lib().grlhip_sample_stages_backward(self.handle, _ptr(packed), L.F32, _ptr(states), _ptr(cond),
                                                          _ptr(table), n1 - 1, rows, _ptr(grad_states), _ptr(gx0),
                                                          _ptr(flat), _ptr(ws), ws.numel(), _stream(dev)))
        return flat, gx0

    def energy_grad(self, packed, t, x, cond, scale, dtype=L.F32):
        rows = x.shape[0]
        dev = x.device
        t, x, cond = _f32c(t, "t", (rows,)), _f32c(x, "x", (rows, self.x_dim)), _f32c(cond, "condition")
        out = torch.empty(rows, self.x_dim, dtype=torch.float32, device=dev)
        ws = self.workspace(rows, dev, None, dtype)
        with torch.cuda.device(dev):
            L.check(L.lib().grlhip_energy_grad(self.handle, _ptr(packed), dtype, _ptr(t), _ptr(x), _ptr(cond),
                                               float(scale), _ptr(out), rows, _ptr(ws), ws.numel(), _stream(dev)))
        return out

    def sample_stages(self, packed, x0, cond, table, guidance=None, guidance_packed=None, trajectory_slots=0,
                      dtype=L.F32):
        """table: (n_stages, COEF_STRIDE) fp32 stage table (include/grlhip.h, grlhip_sample_stages).
        Returns (final, traj|None); traj slot 0 is x0."""
        rows, D = x0.shape
        dev = x0.device
        x0, cond = _f32c(x0, "x_0", (rows, self.x_dim)), _f32c(cond, "condition")
        table = _f32c(table.to(dev), "stage table")
        assert table.dim() == 2 and table.shape[1] == L.COEF_STRIDE
        final = torch.empty(rows, D, dtype=torch.float32, device=dev)
        traj = torch.empty(trajectory_slots, rows, D, dtype=torch.float32, device=dev) if trajectory_slots else None
        ws = self.workspace(rows, dev, guidance, dtype)
        g_h = guidance.handle if guidance is not None else None
        with torch.cuda.device(dev):
            L.check(L.lib().grlhip_sample_stages(self.handle, _ptr(packed), dtype, g_h, _ptr(guidance_packed), _ptr(x0),
                                                 _ptr(cond), _ptr(table), table.shape[0], rows, _ptr(final), _ptr(traj),
                                                 _ptr(ws), ws.numel(), _stream(dev)))
        return final, traj

    def sample(self, packed, x0, cond, coef, dW=None, guidance=None, guidance_packed=None, trajectory_slots=0,
               dtype=L.F32):
        """coef: (n_steps, COEF_STRIDE) fp32 table (see include/grlhip.h).  Returns (final, traj|None)."""
        rows, D = x0.shape
        dev = x0.device
        x0, cond = _f32c(x0, "x_0", (rows, self.x_dim)), _f32c(cond, "condition")
        coef = _f32c(coef.to(dev), "coef")
        n_steps = coef.shape[0]
        assert coef.dim() == 2 and coef.shape[1] == L.COEF_STRIDE
        dW = _f32c(dW, "noise increments", (n_steps, rows, D)) if dW is not None else None
        final = torch.empty(rows, D, dtype=torch.float32, device=dev)
        traj = torch.empty(trajectory_slots, rows, D, dtype=torch.float32, device=dev) if trajectory_slots else None
        ws = self.workspace(rows, dev, guidance, dtype)
        g_h = guidance.handle if guidance is not None else None
        with torch.cuda.device(dev):
            if dW is None:
                L.check(L.lib().grlhip_sample_ode(self.handle, _ptr(packed), dtype, g_h, _ptr(guidance_packed),
                                                  _ptr(x0), _ptr(cond), _ptr(coef), n_steps, rows, _ptr(final),
                                                  _ptr(traj), _ptr(ws), ws.numel(), _stream(dev)))
            else:
                L.check(L.lib().grlhip_sample_sde(self.handle, _ptr(packed), dtype, g_h, _ptr(guidance_packed),
                                                  _ptr(x0), _ptr(cond), _ptr(coef), _ptr(dW), n_steps, rows,
                                                  _ptr(final), _ptr(traj), _ptr(ws), ws.numel(), _stream(dev)))
        return final, traj


    def sample_fixed(self, packed, x0, cond, coef, fixed_x, fixed_mask, fix_sc, fix_noise=None, dW=None, guidance=None,
                     guidance_packed=None, trajectory_slots=0, dtype=L.F32):
        """Sampling with pinned components (include/grlhip.h, grlhip_sample_fixed).  fixed_x / fixed_mask: (rows, D);
        fix_sc: (n_steps, 2) = scale / std of the forward path at each step's network time; fix_noise:
        (n_steps, rows, D) or None.  Returns (final, traj|None); traj slot 0 is the blended initial state."""
        rows, D = x0.shape
        dev = x0.device
        x0, cond = _f32c(x0, "x_0", (rows, self.x_dim)), _f32c(cond, "condition")
        coef = _f32c(coef.to(dev), "coef")
        n_steps = coef.shape[0]
        assert coef.dim() == 2 and coef.shape[1] == L.COEF_STRIDE
        fixed_x, fixed_mask = _f32c(fixed_x, "fixed_x", (rows, D)), _f32c(fixed_mask, "fixed_mask", (rows, D))
        fix_sc = _f32c(fix_sc.to(dev), "fix_sc", (n_steps, 2))
        fix_noise = _f32c(fix_noise, "fixed noise", (n_steps, rows, D)) if fix_noise is not None else None
        dW = _f32c(dW, "noise increments", (n_steps, rows, D)) if dW is not None else None
        final = torch.empty(rows, D, dtype=torch.float32, device=dev)
        traj = torch.empty(trajectory_slots, rows, D, dtype=torch.float32, device=dev) if trajectory_slots else None
        ws = self.workspace(rows, dev, guidance, dtype)
        g_h = guidance.handle if guidance is not None else None
        with torch.cuda.device(dev):
            L.check(L.lib().grlhip_sample_fixed(self.handle, _ptr(packed), dtype, g_h, _ptr(guidance_packed), _ptr(x0),
                                                _ptr(cond), _ptr(coef), _ptr(dW), n_steps, rows, _ptr(fixed_x),
                                                _ptr(fixed_mask), _ptr(fix_sc), _ptr(fix_noise), _ptr(final), _ptr(traj),
                                                _ptr(ws), ws.numel(), _stream(dev)))
        return final, traj


class _DifferentiableForward(torch.autograd.Function):
    """net(t, x, cond) with a backward to the parameters and to x (grlhip_net_backward); no gradient to t / cond."""

    @staticmethod
    def forward(ctx, owner, t, x, cond, *params):
        ctx.owner = owner
        ctx.save_for_backward(*[v for v in (t, x, cond) if v is not None])
        ctx.has = (t is not None, cond is not None)
        return owner.plan.forward(owner.packed(L.F32), t, x, cond, L.F32)

    @staticmethod
    def backward(ctx, gout):
        owner = ctx.owner
        saved = list(ctx.saved_tensors)
        t = saved.pop(0) if ctx.has[0] else None
        x = saved.pop(0)
        cond = saved.pop(0) if ctx.has[1] else None
        plan = owner.plan
        named = owner._named()
        need_params = any(ctx.needs_input_grad[4:])
        need_x = ctx.needs_input_grad[2]
        flat, gx = plan.backward(owner.packed(L.F32), t, x, cond, gout, need_params, need_x)
        grads = []
        for i, (name, shape) in enumerate(plan.params):
            p = named[name]
            if p.requires_grad and flat is not None:
                off = L.lib().grlhip_grad_offset(plan.handle, i)
                grads.append(flat[off:off + p.numel()].view(p.shape))
            else:
                grads.append(None)
        return (None, None, gx, None, *grads)


def differentiable_forward(owner, t, x, cond):
    """owner: IntrinsicModel / QNetwork (``plan``, ``packed()``, ``_named()``).  Plain forward when autograd is off or
    neither x nor a parameter requires grad; otherwise the output carries a grad_fn whose backward reaches the
    parameters and x (for a Q network: the action)."""
    named = owner._named()
    params = [named[n] for n, _ in owner.plan.params]
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
        return _DifferentiableForward.apply(owner, t, x, cond, *params)
    return owner.plan.forward(owner.packed(L.F32), t, x, cond, L.F32)
