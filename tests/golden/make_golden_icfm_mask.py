"""Golden vectors for the masked ICFM entry points of the reference (used by grl/algorithms/gmpg_mask.py):
``flow_matching_loss_with_mask`` (conditional_flow_model/independent_conditional_flow_model.py:699-741, loss + every
parameter gradient), the element-wise ``flow_matching_loss(average=False, sum_all_elements=False)`` underneath it
(model_functions/velocity_function.py:248-253) and ``sample_with_mask(_forward_process)`` (:121-168, 452-513).
Container-only script:   python tests/golden/make_golden_icfm_mask.py   ->  tests/golden/icfm_mask.npz
The random draws (t = rand(B) * t_max, then randn_like inside direct_sample) are captured by replaying the generator."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import golden_lib as gl  # noqa: E402
from ref_shims import EasyDict  # noqa: E402
from grl.generative_models.conditional_flow_model.independent_conditional_flow_model import (  # noqa: E402
    IndependentConditionalFlowModel,
)

torch.set_num_threads(8)
out = {}
for name, rows in (("small_icfm", 40), ("icfm_cond", 24)):
    kind, cfg = gl.model_zoo()[name]
    cfg = EasyDict(cfg)
    m = IndependentConditionalFlowModel(cfg)
    key_shapes = [(k, list(v.shape)) for k, v in m.state_dict().items()]
    m.load_state_dict({k: torch.from_numpy(v) for k, v in gl.load_weights(name, key_shapes).items()})
    D, S = cfg.x_size, cfg.model.args.backbone.args.condition_dim
    gen = torch.Generator().manual_seed(gl._seed_for(name, 66))
    x0, x1 = torch.randn(rows, D, generator=gen), torch.randn(rows, D, generator=gen) * 0.8
    cond = torch.randn(rows, S, generator=gen)
    mask = torch.rand(rows, D, generator=gen) < 0.6
    mask[0, :] = False   # a row with nothing kept: divided by clamp(0, min=1)
    mask[1, :] = True
    pre = name + "/"
    out[pre + "x0"], out[pre + "x1"], out[pre + "cond"], out[pre + "mask"] = x0.numpy(), x1.numpy(), cond.numpy(), mask.numpy()
    seed = gl._seed_for(name, 67)
    torch.manual_seed(seed)
    out[pre + "t"] = (torch.rand(rows) * 1.0).numpy()
    out[pre + "eps"] = torch.randn(rows, D).numpy()
    for p in m.parameters():
        p.grad = None
    torch.manual_seed(seed)
    loss = m.flow_matching_loss_with_mask(x0, x1, cond, mask, average=True)
    loss.backward()
    out[pre + "loss"] = loss.detach().numpy()
    for k, p in m.named_parameters():
        if p.grad is not None:
            g = p.grad.detach().numpy().astype(np.float32)
            if name == "small_icfm":
                out[pre + "grad/" + k] = g
            else:
                flat = g.reshape(-1)
                out[pre + "gradsum/" + k] = np.asarray([flat.astype(np.float64).sum(), np.sqrt((flat.astype(np.float64) ** 2).sum())])
    torch.manual_seed(seed)
    out[pre + "loss_rows"] = m.flow_matching_loss_with_mask(x0, x1, cond, mask, average=False).detach().numpy()
    torch.manual_seed(seed)
    out[pre + "loss_elements"] = m.velocity_function_.flow_matching_loss_icfm(
        m.model, x0, x1, cond, average=False, sum_all_elements=False).detach().numpy()
    # sampling with a mask: the elements under the mask come from x_1
    t_span = torch.linspace(0.0, 1.0, 6)
    with torch.no_grad():
        traj = m.sample_with_mask_forward_process(t_span=t_span, x_0=x0, condition=cond, mask=mask, x_1=x1)
        final = m.sample_with_mask(t_span=t_span, x_0=x0, condition=cond, mask=mask, x_1=x1)
    assert torch.equal(final, traj[-1])
    out[pre + "sample_traj"] = traj.numpy()
    print(name, float(loss), tuple(traj.shape))
np.savez_compressed(os.path.join(HERE, "icfm_mask.npz"), **out)
print("wrote icfm_mask.npz", sum(a.nbytes for a in out.values()) // 1024, "KiB")
