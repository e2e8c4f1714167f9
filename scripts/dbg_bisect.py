"""Hardware bisect of the bf16 integrator by crafted weights: modules are enabled cumulatively (all others zero) and the
one-step bf16 result is compared with the fp32 engine."""
import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tests/golden')
import golden_lib as gl, golden_util as gu
import generativerl_preview_amd as G
from generativerl_preview_amd import _lib as L
name = sys.argv[1] if len(sys.argv) > 1 else "qgpo_noise"
kind, cfg = gl.model_zoo()[name]
cfg = dict(cfg, device="cuda:0"); cfg.pop("energy_guidance", None)
full = {k: v for k, v in gu.weights(name).items() if k.startswith("model.")}
order = ["last_block.bias", "first_block.modify_x.bias", "last_block.weight", "first_block.modify_x.weight",
         "first_block.dense2.0.bias", "first_block.dense1.0.bias", "first_block.dense2.0.weight", "first_block.time_mlp.1.bias",
         "first_block.dense1.0.weight", "sort_t.2.bias", "first_block.time_mlp.1.weight", "sort_t.0.bias", "sort_t.2.weight",
         "sort_t.0.weight", "pre_sort_condition", "down_block.0.modify_x", "down_block.0.dense2", "down_block.0.dense1", "down_block.0.time_mlp",
         "down_block.1.modify_x", "down_block.1.dense2", "down_block.1.dense1", "down_block.1.time_mlp",
         "middle_block.dense2", "middle_block.dense1", "middle_block.time_mlp",
         "up_block.0.dense2", "up_block.0.dense1", "up_block.0.time_mlp", "up_block.1.dense2", "up_block.1.dense1", "up_block.1.time_mlp"]
torch.manual_seed(0)
D = cfg["x_size"]
rows = 64
x0 = torch.randn(rows, D, device="cuda")
kw = dict(x_0=x0)
if D == 6:
    kw["condition"] = torch.randn(rows, 17, device="cuda")
ts = torch.linspace(0, 1, 2)
enabled = []
for stage in [None] + order:
    if stage is not None:
        enabled.append(stage)
    sd = {}
    for k, v in full.items():
        on = k.endswith("t_encoder.W") or any(e in k for e in enabled)
        sd[k] = v.clone() if on else torch.zeros_like(v)
    m = G.DiffusionModel(cfg); m.load_state_dict(sd); m = m.to("cuda:0")
    m.model.compute_dtype = L.F32
    ref = m.sample(t_span=ts, **kw)
    m.model.compute_dtype = L.BF16
    got = m.sample(t_span=ts, **kw)
    err = (got - ref).abs()
    print("%-36s max err %.3e  nan %d  max|ref - x0| %.3e" % (stage, float(err[~torch.isnan(err)].max()), int(torch.isnan(got).sum()), float((ref - x0).abs().max())), flush=True)
