import sys, time, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tests/golden')
import util_models as um
kind,cfg,m = um.build("qgpo_noise","cuda:0")
for rows in (32768, 262144):
    x0=torch.randn(rows,6,device="cuda"); c=torch.randn(rows,17,device="cuda")
    ts=torch.linspace(0,1,101)
    for gs in (None,):
        m2 = m
        # unguided path: plain DiffusionModel semantics via guidance 0 is still guided; time both
        torch.cuda.synchronize(); t0=time.time()
        out=m.sample(t_span=ts,x_0=x0,condition=c,guidance_scale=0.0); torch.cuda.synchronize(); t1=time.time()
        print("guided rows",rows,"time",t1-t0,"action-steps/s",rows*100/(t1-t0))
import generativerl_preview_amd as G, golden_lib as gl
cfg2=dict(gl.diffusion_cfg(6, gl.VP, "noise_function", [512,256,128], 17), device="cuda:0")
dm=G.DiffusionModel(cfg2).to("cuda:0")
for rows in (32768, 262144):
    x0=torch.randn(rows,6,device="cuda"); c=torch.randn(rows,17,device="cuda")
    for i in range(2):
        torch.cuda.synchronize(); t0=time.time()
        out=dm.sample(t_span=ts,x_0=x0,condition=c); torch.cuda.synchronize(); t1=time.time()
        print("unguided rows",rows,"time",t1-t0,"action-steps/s",rows*100/(t1-t0), "TFLOP/s", rows*100*3.253e6/(t1-t0)/1e12)
