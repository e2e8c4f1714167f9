"""C5 (GVP velocity net [1024], D = 64, 100-step Euler ODE, 262 144 rows) on the bf16 batch engine: seconds per
sample() call and TFLOP/s (the same measurement as bench.py's C5_gvp_wide secondary)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import generativerl_preview_amd.generative_models as G  # noqa: E402
from generativerl_preview_amd import _lib as L  # noqa: E402

device = "cuda:0"
t_enc = dict(type="GaussianFourierProjectionTimeEncoder", args=dict(embed_dim=32, scale=30.0))
cfg = dict(device=device, x_size=64, alpha=1.0, path=dict(type="gvp"),
           solver=dict(type="ODESolver", args=dict(library="torchdyn")),
           model=dict(type="velocity_function", args=dict(t_encoder=t_enc, backbone=dict(
               type="TemporalSpatialResidualNet", args=dict(hidden_sizes=[1024], output_dim=64, t_dim=32)))))
torch.manual_seed(0)
m5 = G.DiffusionModel(cfg).to(device)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
x0 = torch.randn(rows, 64, device=device)
t_span = torch.linspace(0.0, 1.0, 101, device=device)
only = sys.argv[2] if len(sys.argv) > 2 else ""       # "bf16" / "f32": that engine only
n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
t_span = torch.linspace(0.0, 1.0, n_steps + 1, device=device)
for dtype, name in ((L.BF16, "bf16"), (L.F32, "f32")):
    if only and name != only:
        continue
    m5.model.compute_dtype = dtype
    m5.sample(t_span=t_span, x_0=x0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m5.sample(t_span=t_span, x_0=x0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("C5 %s: %.3f s per call, %.3e action-steps/s, %.1f TFLOP/s" % (name, dt, rows * n_steps / dt, rows * n_steps * 6950912 / dt / 1e12))
