"""Adam over ONE flat parameter buffer (grlhip_adam_step): the optimizer of the reference's training loops
(``torch.optim.Adam`` over one network's parameters, grl/algorithms/qgpo.py:554,637,642, gmpo.py:788,942,946,1061) as a
single launch instead of the per-tensor kernels of the foreach implementation.

``FlatAdam(owner)`` takes an IntrinsicModel-like owner (``plan``, ``_named()``; what ``attach_flat_grads`` takes).  The
parameters are moved into ONE flat fp32 buffer in the library's gradient layout (their ``.data`` become views of it, so
``state_dict()``, ``named_parameters()`` and the weight packing see ordinary tensors) and their ``.grad`` are the views
of the FlatGrads buffer the fused losses accumulate into.  ``step()`` is one kernel over the trainable range(s);
``zero_grad()`` one memset.  Same update rule and defaults as ``torch.optim.Adam`` (amsgrad / maximize off).

``capturable=True`` keeps the step count on the device (grlhip_adam_step_counted), as ``torch.optim.Adam(capturable=True)``
does: a whole training step -- loss forward + backward, gradient all-reduce, ``step()`` -- can then be captured once with
``torch.cuda.graph`` and replayed (``tests/test_optim.py``)."""
import torch

from . import _lib as L
from .distributed import attach_flat_grads


class FlatAdam:
    def __init__(self, owner, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError("invalid Adam hyper-parameters")
        self.owner = owner
        self.fg = attach_flat_grads(owner)
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        self.param_groups = [dict(self.defaults, params=[p for p, _, _ in self.fg.slices if p.requires_grad])]
        self._steps = 0
        self.capturable = bool(capturable)
        dev = self.fg.flat.device
        # capturable: {int64 step; float step_size; float bc2_sqrt} on the device, advanced by the library
        self.step_record = torch.zeros(2, dtype=torch.int64, device=dev) if capturable else None
        n = self.fg.n
        self.flat_params = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, off, cnt in self.fg.slices:
                view = self.flat_params[off:off + cnt].view(p.shape)
                view.copy_(p.data)
                p.data = view                      # the module's parameter now lives in the flat buffer
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        # contiguous ranges of trainable elements (frozen parameters, e.g. the Fourier frequencies, are skipped)
        self.ranges = []
        for p, off, cnt in sorted(self.fg.slices, key=lambda s: s[1]):
            if not p.requires_grad:
                continue
            if self.ranges and self.ranges[-1][1] == off:
                self.ranges[-1][1] = off + cnt
            else:
                self.ranges.append([off, off + cnt])

    def _check_views(self):
        base = self.flat_params.data_ptr()
        for p, off, cnt in self.fg.slices:
            if p.data_ptr() != base + 4 * off:
                raise RuntimeError("a parameter no longer lives in FlatAdam's buffer (module moved or re-assigned): "
                                   "build a new FlatAdam")

    @property
    def step_count(self):
        return int(self.step_record[0].item()) if self.capturable else self._steps

    @step_count.setter
    def step_count(self, value):
        if self.capturable:
            self.step_record.zero_()
            self.step_record[0] = int(value)
        else:
            self._steps = int(value)

    def zero_grad(self, set_to_none=False):
        self.fg.ensure()
        self.fg.zero()

    @torch.no_grad()
    def step(self):
        self._check_views()
        self.fg.ensure()
        g = self.param_groups[0]
        dev = self.flat_params.device
        if not self.capturable:
            self._steps += 1
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            for i, (lo, hi) in enumerate(self.ranges):
                ptrs = (self.flat_params.data_ptr() + 4 * lo, self.fg.flat.data_ptr() + 4 * lo,
                        self.exp_avg.data_ptr() + 4 * lo, self.exp_avg_sq.data_ptr() + 4 * lo, hi - lo,
                        g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"])
                if self.capturable:   # the step count lives on the device: nothing per-step is frozen into a captured graph
                    L.check(L.lib().grlhip_adam_step_counted(*ptrs, self.step_record.data_ptr(), int(i == 0), stream))
                else:
                    L.check(L.lib().grlhip_adam_step(*ptrs, self._steps, stream))
        # the library wrote the parameters behind torch's back: their version counters did not move, and the owner's
        # packed-image cache (IntrinsicModel.packed) is keyed on them -- drop it explicitly
        if hasattr(self.owner, "_packed"):
            self.owner._packed = {} if isinstance(self.owner._packed, dict) else None

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.param_groups[0].update(sd["param_groups"][0])
