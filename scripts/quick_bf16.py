import sys, time, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tests/golden')
import golden_lib as gl
import generativerl_preview_amd as G
from generativerl_preview_amd import _lib as L
cfg=dict(gl.diffusion_cfg(6, gl.VP, "noise_function", [512,256,128], 17), device="cuda:0")
dm=G.DiffusionModel(cfg).to("cuda:0")
dm.model.compute_dtype = L.BF16
ts=torch.linspace(0,1,101)
for rows in (32768, 262144, 1048576):
    x0=torch.randn(rows,6,device="cuda"); c=torch.randn(rows,17,device="cuda")
    for i in range(2):
        torch.cuda.synchronize(); t0=time.time()
        out=dm.sample(t_span=ts,x_0=x0,condition=c); torch.cuda.synchronize(); t1=time.time()
        print("bf16 rows",rows,"time %.4f"%(t1-t0),"action-steps/s %.3e"%(rows*100/(t1-t0)), "TFLOP/s %.1f"%(rows*100*3.253312e6/(t1-t0)/1e12), flush=True)
