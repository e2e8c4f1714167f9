"""Per-launch duration of the bench kernel over consecutive launches on a fresh box (clock / power behaviour)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from generativerl_preview_amd import engine
model = bench.build_model("cuda:0")
states = torch.randn(bench.N_STATES, 17, device="cuda:0")
t_span = torch.linspace(0.0, 1.0, bench.N_SOLVER_STEPS + 1)
engine.KERNEL_EVENTS = []
for _ in range(5):
    model.sample(t_span=t_span, batch_size=bench.N_CAND, condition=states)
torch.cuda.synchronize()
print(json.dumps({"kernel_ms_per_launch": [round(a.elapsed_time(b), 1) for a, b in engine.KERNEL_EVENTS]}))
