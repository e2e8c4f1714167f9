import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tests/golden')
import golden_lib as gl, golden_util as gu
import generativerl_preview_amd as G
from generativerl_preview_amd import _lib as L
name = "qgpo_noise"
kind, cfg = gl.model_zoo()[name]
cfg = dict(cfg, device="cuda:0"); cfg.pop("energy_guidance", None)
full = {k: v for k, v in gu.weights(name).items() if k.startswith("model.")}
enabled = ["last_block.bias", "first_block.modify_x.bias", "last_block.weight", "first_block.modify_x.weight",
         "first_block.dense2.0.bias", "first_block.dense1.0.bias", "first_block.dense2.0.weight", "first_block.time_mlp.1.bias",
         "first_block.dense1.0.weight"]
torch.manual_seed(0)
rows = 64
cond = torch.randn(rows, 17, device="cuda")
ts = torch.linspace(0, 1, 2)
def run(x0, w1scale=1.0, only=None):
    sd = {}
    for k, v in full.items():
        on = k.endswith("t_encoder.W") or any(e in k for e in (only or enabled))
        sd[k] = v.clone() if on else torch.zeros_like(v)
        if "first_block.dense1.0.weight" in k: sd[k] = sd[k] * w1scale
    m = G.DiffusionModel(cfg); m.load_state_dict(sd); m = m.to("cuda:0")
    m.model.compute_dtype = L.F32; ref = m.sample(t_span=ts, x_0=x0, condition=cond)
    m.model.compute_dtype = L.BF16; got = m.sample(t_span=ts, x_0=x0, condition=cond)
    return got, ref
x0 = torch.randn(rows, 6, device="cuda")
for label, args in (("x0=0", dict(x0=torch.zeros_like(x0))), ("w1 x 0.01", dict(x0=x0, w1scale=0.01)), ("w1 x 0.1", dict(x0=x0, w1scale=0.1)), ("full", dict(x0=x0)),
                    ("only dense1 w+b, dense2 w, last", dict(x0=x0, only=["last_block", "first_block.dense1.0", "first_block.dense2.0.weight"]))):
    got, ref = run(**args)
    e = (got - ref).abs()
    print(label, "max err %.3e" % float(e.max()), "rows with err>0.5:", int((e.max(1).values > 0.5).sum()), "worst row", int(e.max(1).values.argmax()))
    r = int(e.max(1).values.argmax())
    print("   got", [round(v, 3) for v in got[r].tolist()], "ref", [round(v, 3) for v in ref[r].tolist()])
    print("   err per row (first 40):", [round(float(v), 2) for v in e.max(1).values[:40]])
