import sys, time, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests/golden')
import golden_lib as gl
import generativerl_preview_amd as G
cfg=dict(gl.icfm_cfg(6,[512,256,128],17), device="cuda:0")
m=G.IndependentConditionalFlowModel(cfg).to("cuda:0")
opt=torch.optim.Adam(m.parameters(), lr=1e-4)
for B in (8192, 65536):
    x1=torch.rand(B,6,device="cuda")*2-1; c=torch.randn(B,17,device="cuda")
    for i in range(3):
        torch.cuda.synchronize(); t0=time.time()
        opt.zero_grad(); x0=torch.randn(B,6,device="cuda")
        loss=m.flow_matching_loss(x0,x1,c); loss.backward(); opt.step()
        torch.cuda.synchronize(); dt=time.time()-t0
        print("B",B,"step %.4f s"%dt,"rows/s %.3e"%(B/dt),"TFLOP/s %.2f"%(B*9.759936e6/dt/1e12),"loss %.4f"%float(loss), flush=True)
