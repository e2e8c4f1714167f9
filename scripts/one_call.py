"""One bf16 sampling call (524288 rows x 100 steps) -- target for rocprofv3 PMC passes."""
import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests/golden')
import golden_lib as gl
import generativerl_preview_amd as G
from generativerl_preview_amd import _lib as L
cfg = dict(gl.diffusion_cfg(6, gl.VP, "noise_function", [512, 256, 128], 17), device="cuda:0")
dm = G.DiffusionModel(cfg).to("cuda:0")
dm.model.compute_dtype = L.BF16
rows = 524288
x0 = torch.randn(rows, 6, device="cuda"); c = torch.randn(rows, 17, device="cuda")
out = dm.sample(t_span=torch.linspace(0, 1, 101), x_0=x0, condition=c)
torch.cuda.synchronize()
print("ok", float(out.abs().mean()))
