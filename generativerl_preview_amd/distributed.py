"""Batch sharding across the GPUs of one node (one process per GPU, torch.distributed over RCCL).

The reference is single-process; this is the minimal data-parallel layer of
the hot path: sampling shards states in contiguous blocks and needs NO collective; training sums
the per-rank gradients with ONE all-reduce of a single flat fp32 buffer per step."""
import torch
import torch.distributed as dist


def shard_rows(n_rows: int, rank: int = None, world_size: int = None):
    """Contiguous block [lo, hi) of `n_rows` owned by `rank` (blocks differ by at most one row), so
    that the K candidates of one state stay on one GPU when rows are states."""
    rank = dist.get_rank() if rank is None else rank
    world_size = dist.get_world_size() if world_size is None else world_size
    base, rem = divmod(n_rows, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class FlatGrads:
    """ONE persistent flat fp32 gradient buffer for an IntrinsicModel-like owner (``plan``, ``_named()``): the
    library's flat gradient layout (grlhip_grad_offset) plus two tail slots [loss * rows, rows].  The parameters'
    ``.grad`` are VIEWS of the buffer, the fused loss accumulates into it in one pass (model_functions/losses.py),
    optimizers and ``clip_grad_norm_`` see ordinary ``.grad`` tensors, and the data-parallel reduction is one
    in-place all-reduce of this buffer -- no per-parameter copies, no ``cat``, no copy-back.

    ``torch.optim.Optimizer.zero_grad()`` (set_to_none=True) drops the views; ``ensure()`` re-installs them (and zeroes
    the buffer) before the next accumulation, so either zero_grad flavour works."""

    def __init__(self, owner, device=None):
        from . import _lib as L
        plan = owner.plan
        named = owner._named()
        self.owner = owner
        self.n = int(L.lib().grlhip_grad_numel(plan.handle))
        dev = device if device is not None else next(iter(named.values())).device
        self.flat = torch.zeros(self.n + 2, dtype=torch.float32, device=dev)
        self.slices = []
        for i, (name, shape) in enumerate(plan.params):
            off = int(L.lib().grlhip_grad_offset(plan.handle, i))
            p = named[name]
            self.slices.append((p, off, p.numel()))
        self.ensure()

    @property
    def grads(self):
        return self.flat[:self.n]

    def _view(self, p, off, n):
        return self.flat[off:off + n].view(p.shape)

    def ensure(self):
        """(Re-)install the .grad views.  Parameters whose .grad was dropped or replaced get their slice zeroed."""
        stale = [(p, off, n) for p, off, n in self.slices
                 if p.requires_grad and (p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * off)]
        if not stale:
            return
        if len(stale) == sum(1 for p, _, _ in self.slices if p.requires_grad):
            self.flat.zero_()            # the usual case after zero_grad(set_to_none=True): one memset
        for p, off, n in stale:
            v = self._view(p, off, n)
            if len(stale) != len(self.slices):
                v.zero_()
            p.grad = v

    def zero(self):
        self.flat.zero_()

    def clip_grad_norm_(self, max_norm: float, norm_type: float = 2.0):
        """``torch.nn.utils.clip_grad_norm_`` over the owner's parameters (the swiss-roll training loop of the reference,
        grl_pipelines/tutorials: ``clip_grad_norm_(model.parameters(), 1.0)``) on the ONE flat buffer their ``.grad``
        alias: a norm, a clamp and a multiply instead of one norm per tensor -- same formula (``max_norm / (total +
        1e-6)``, clamped to 1), no host synchronisation.  Returns the total norm (0-dim tensor)."""
        g = self.grads
        total = torch.linalg.vector_norm(g, float(norm_type))
        g.mul_(torch.clamp(float(max_norm) / (total + 1e-6), max=1.0))
        return total

    def allreduce(self, local_rows: int, local_loss=None, group=None):
        """Turn per-rank gradients of a mean over the LOCAL rows into the gradient of the mean over the GLOBAL batch
        (the reference's ``mean`` over all rows): weight by the local row count, ONE all-reduce(SUM) of the flat buffer
        (gradients, loss and row count travel together), divide by the global row count.  Ranks may hold different
        numbers of rows.  Returns the global mean loss (a 0-dim tensor) when ``local_loss`` is given."""
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            return local_loss
        g = self.flat
        g[:self.n] *= float(local_rows)
        g[self.n] = 0.0 if local_loss is None else local_loss.detach() * float(local_rows)
        g[self.n + 1] = float(local_rows)
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
        g[:self.n + 1] /= g[self.n + 1]
        return g[self.n].clone() if local_loss is not None else None


def attach_flat_grads(owner, device=None) -> FlatGrads:
    """Give `owner` (an IntrinsicModel / QNetwork) a FlatGrads buffer; the fused losses then accumulate straight
    into it.  Idempotent."""
    fg = getattr(owner, "_flat_grads", None)
    if fg is None:
        fg = FlatGrads(owner, device)
        owner._flat_grads = fg
    return fg


def allreduce_gradients(parameters, group=None, average: bool = True):
    """Sum (or average) `.grad` of `parameters` over the process group with a single all-reduce of
    one flat buffer (1 633 222 trainable fp32 values = 6.5 MB for the QGPO/GMPO net).  With
    per-rank losses that are means over the LOCAL batch, average=True reproduces the reference's
    mean over the global batch when every rank holds the same number of rows."""
    params = [p for p in parameters if p.grad is not None]
    if not params or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    # generic fallback for arbitrary parameter lists (critics trained through torch autograd); models with an attached
    # FlatGrads buffer use FlatGrads.allreduce, which needs neither the cat nor the copy-back below
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n
