"""Loss and a hash of every parameter gradient of five test nets on fixed inputs: run it once per library build
(``GRLHIP_LIB=/path/to/libgrlhip_<name>.so python scripts/ab_grad_hash.py [backward_split]``, see build_variant.sh) and
compare the lines -- a kernel or scheduling change that is meant to keep the arithmetic must keep every hash."""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import util_models as um  # noqa: E402

DEV = torch.device("cuda:0")
split = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for name, B in (("icfm_cond", 20000), ("small_icfm", 65536 + 333), ("qgpo_noise", 4096), ("gvp_wide", 3000), ("swiss_noise", 700)):
    kind, cfg, m = um.build(name, DEV)
    m.model.backward_split = split
    gen = torch.Generator().manual_seed(5)
    xd, cd = m.model.plan.x_dim, m.model.plan.cond_dim
    x0, x1 = torch.randn(B, xd, generator=gen).to(DEV), torch.rand(B, xd, generator=gen).to(DEV)
    cond = torch.randn(B, cd, generator=gen).to(DEV) if cd else None
    t, eps = torch.rand(B, generator=gen).to(DEV), torch.randn(B, xd, generator=gen).to(DEV)
    m.zero_grad(set_to_none=True)
    if kind == "icfm":
        loss = m.flow_matching_loss(x0, x1, cond, t=t, noise=eps)
    else:
        loss = m.score_matching_loss(x1, cond, weighting_scheme="vanilla", t=t * 0.99 + 0.005, noise=eps)
    loss.backward()
    h = hashlib.sha256()
    for k, p in sorted(m.named_parameters()):
        if p.grad is not None:
            h.update(p.grad.detach().cpu().numpy().tobytes())
    print("split %d %-12s rows %6d loss %.9g grads %s" % (split, name, B, float(loss), h.hexdigest()[:16]))
