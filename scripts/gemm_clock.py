"""Shader clock and cycle split of k_bt_gemm_x6 INSIDE the C3 training step (real activations, not the microbenchmark's
random data): needs a library built with -DGEMM_STAMPS, loaded through GRLHIP_LIB (see the build line below); the
kernel sums s_memtime (shader cycles) and s_memrealtime (100 MHz) per workgroup.

    T=$(mktemp -d); C=generativerl_preview_amd/csrc
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -mllvm -pragma-unroll-threshold=1048576 -DGEMM_STAMPS -Iinclude -c $C/grlhip.hip -o $T/g.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o generativerl_preview_amd/libgrlhip_stamps.so $T/g.o $(ls $C/build/*.o | grep -v grlhip.o)
    GRLHIP_LIB=$PWD/generativerl_preview_amd/libgrlhip_stamps.so python scripts/gemm_clock.py [backward_split]
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import golden_lib as gl  # noqa: E402
import generativerl_preview_amd as G  # noqa: E402
from generativerl_preview_amd import _lib as L  # noqa: E402
from generativerl_preview_amd.distributed import attach_flat_grads  # noqa: E402
from generativerl_preview_amd.optim import FlatAdam  # noqa: E402

dev = torch.device("cuda:0")
split = int(sys.argv[1]) if len(sys.argv) > 1 else 3
torch.manual_seed(0)
m = G.IndependentConditionalFlowModel(dict(gl.icfm_cfg(6, [512, 256, 128], 17), device=dev)).to(dev)
m.model.backward_split = split
fg = attach_flat_grads(m.model)
opt = FlatAdam(m.model, lr=1e-4)
B = 65536
x1 = torch.rand(B, 6, device=dev) * 2 - 1
cond = torch.randn(B, 17, device=dev)


def step():
    fg.zero()
    m.flow_matching_loss(torch.randn(B, 6, device=dev), x1, cond).backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
dbg = ctypes.CDLL(L.LIB_PATH).grlhip_dbg_gemm   # (same handle as the package's: dlopen of a loaded library)
dbg.argtypes = [ctypes.c_void_p, ctypes.c_int]
dbg(None, 1)
n = 10
for _ in range(n):
    step()
torch.cuda.synchronize()
z = (ctypes.c_ulonglong * 8)()
dbg(z, 0)
wg = float(z[6])
print("backward split %d, %d steps: %.0f GEMM workgroups per step; per workgroup: prologue %.0f, k-loop %.0f (barrier wait %.0f, "
      "load wait %.0f), epilogue %.0f, whole %.0f cycles = %.2f us; shader clock %.2f GHz"
      % (split, n, wg / n, z[0] / wg, z[1] / wg, z[2] / wg, z[3] / wg, z[4] / wg, z[5] / wg, z[7] / wg / 100.0,
         z[5] / (z[7] / 100.0) / 1e3))
