import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tests/golden')
import util_models as um, golden_util as gu
from generativerl_preview_amd import _lib as L
torch.manual_seed(0)
for name in ("qgpo_noise", "swiss_noise"):
    kind, cfg, m = um.build(name, "cuda:0")
    D = cfg["x_size"]
    rows = 256
    x0 = torch.randn(rows, D, device="cuda")
    kw = dict(x_0=x0)
    if D == 6:
        kw["condition"] = torch.randn(rows, 17, device="cuda")
    if kind == "energy":
        import generativerl_preview_amd as G
        c2 = dict(cfg); c2.pop("energy_guidance")
        m2 = G.DiffusionModel(c2); m2.load_state_dict({k: v for k, v in gu.weights(name).items() if k.startswith("model.")}); m = m2.to("cuda:0")
    for steps in (1, 2, 10):
        ts = torch.linspace(0, 1, steps + 1)
        m.model.compute_dtype = L.F32
        ref = m.sample(t_span=ts, **kw)
        m.model.compute_dtype = L.BF16
        got = m.sample(t_span=ts, **kw)
        d = (got - ref)
        print(name, "steps", steps, "nan rows", int(torch.isnan(got).any(1).sum()), "max err", float(d[~torch.isnan(d)].abs().max()) if (~torch.isnan(d)).any() else None,
              "max |ref|", float(ref.abs().max()))
        if steps == 1:
            print(" got[0]", got[0].tolist()); print(" ref[0]", ref[0].tolist()); print(" got[33]", got[33].tolist()); print(" ref[33]", ref[33].tolist())
print("---- identical rows")
kind, cfg, m = um.build("swiss_noise", "cuda:0")
x0 = torch.randn(1, 2, device="cuda").repeat(256, 1)
m.model.compute_dtype = L.BF16
got = m.sample(t_span=torch.linspace(0, 1, 2), x_0=x0)
print(got[:8].tolist()); print("distinct rows:", len(set(map(tuple, got.tolist()))))
for w in range(4):
    print("wave", w, got[32*w:32*w+3].tolist())
