"""Timing of experiment builds of libgrlhip (GRLHIP_LIB=...): one process per library, 262144 rows x 100 steps, 3 reps."""
import os, subprocess, sys
libs = sys.argv[1:]
code = r'''
import sys, time, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tests/golden')
import golden_lib as gl
import generativerl_preview_amd as G
from generativerl_preview_amd import _lib as L
cfg=dict(gl.diffusion_cfg(6, gl.VP, "noise_function", [512,256,128], 17), device="cuda:0")
torch.manual_seed(0)
dm=G.DiffusionModel(cfg)
import numpy as np
z=np.load('/root/repo/tests/golden/weights_qgpo_noise.npz')   # the trained weights bench.py times (DVFS is data-dependent)
dm.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(z[k], dtype=np.float32)) for k in z.files if not k.startswith("energy_guidance.")})
dm=dm.to("cuda:0"); dm.model.compute_dtype = L.BF16
ts=torch.linspace(0,1,101); rows=524288
x0=torch.randn(rows,6,device="cuda"); c=torch.randn(rows,17,device="cuda")
best=1e9
for i in range(4):
    torch.cuda.synchronize(); t0=time.time(); out=dm.sample(t_span=ts,x_0=x0,condition=c); torch.cuda.synchronize(); best=min(best,time.time()-t0)
cyc = best / (rows/128/256*100) * 1e6
print("%-28s %.4f s  %.1f TFLOP/s  %.2f us/step/wave" % (sys.argv[1], best, rows*100*3.253312e6/best/1e12, cyc))
'''
for lib in libs:
    env = dict(os.environ, GRLHIP_LIB=os.path.abspath(lib))
    subprocess.run([sys.executable, "-c", code, os.path.basename(lib)], env=env, check=False)
