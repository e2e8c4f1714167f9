"""Phase stamps of the wide GEMM (experiment build with -DWIDE_STAMPS, GRLHIP_LIB=...): cycles wave 0 of every
workgroup spent per phase in the LAST k_w_gemm launch of a C5 sampling call."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import generativerl_preview_amd.generative_models as G
from generativerl_preview_amd import _lib as L
device = "cuda:0"
t_enc = dict(type="GaussianFourierProjectionTimeEncoder", args=dict(embed_dim=32, scale=30.0))
cfg = dict(device=device, x_size=64, alpha=1.0, path=dict(type="gvp"), solver=dict(type="ODESolver", args=dict(library="torchdyn")),
           model=dict(type="velocity_function", args=dict(t_encoder=t_enc, backbone=dict(
               type="TemporalSpatialResidualNet", args=dict(hidden_sizes=[1024], output_dim=64, t_dim=32)))))
torch.manual_seed(0)
m5 = G.DiffusionModel(cfg).to(device)
m5.model.compute_dtype = L.BF16
x0 = torch.randn(262144, 64, device=device)
which = int(sys.argv[1]) if len(sys.argv) > 1 else 0
# run steps, then stop after a chosen layer by monkeypatching? simpler: the last launch of a 1-step call is the output layer;
# env WIDE_STOP (read by nobody) -- instead print stamps of the LAST big launch by running the net forward only up to it.
m5.sample(t_span=torch.linspace(0, 1, 3, device=device), x_0=x0)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (1024 * 8))()
lib = L.lib()
lib.grlhip_dbg_wide.argtypes = [ctypes.c_void_p]
print("rc", lib.grlhip_dbg_wide(buf))
a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.float64)[:256]
names = ["total", "vmcnt wait", "barrier", "k-steps", "epi: stage+barriers", "epi: readback+stores", "epilogue", "tiles"]
for i, n in enumerate(names):
    print("%-18s mean %10.0f  min %10.0f  max %10.0f" % (n, a[:, i].mean(), a[:, i].min(), a[:, i].max()))
