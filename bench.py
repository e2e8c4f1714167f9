#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): action samples/sec for the 100-step Euler ODE of the QGPO
HalfCheetah-shaped policy (obs 17, act 6, TemporalSpatialResidualNet [512,256,128], linear VP-SDE,
noise head), 1 048 576 synthetic states x 32 candidate actions PER GPU, through
``DiffusionModel.sample`` -> C ABI -> the bf16 MFMA persistent integrator.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

One "step" = one full pass (all rows, all 100 solver steps).  Prints ONE JSON line on rank 0.
Sampling shards by state blocks; there is no data-path collective (scaling: weak).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_ACTION_STEP = 3253312          # 2 x 1 626 656 MAC, SURVEY.md section 8d (C2 net)
PEAK_BF16_TFLOPS = 2500.0               # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
N_STATES, N_CAND, N_SOLVER_STEPS = 1048576, 32, 100


def qgpo_config(device):
    """benchmark/qgpo/vpsde/halfcheetah_medium_expert.py:13-17,99-115 (shape only)."""
    t_enc = dict(type="GaussianFourierProjectionTimeEncoder", args=dict(embed_dim=32, scale=30.0))
    return dict(device=device, x_size=6, alpha=1.0,
                path=dict(type="linear_vp_sde", beta_0=0.1, beta_1=20.0),
                solver=dict(type="ODESolver", args=dict(library="torchdyn")),
                model=dict(type="noise_function", args=dict(
                    t_encoder=t_enc,
                    backbone=dict(type="TemporalSpatialResidualNet",
                                  args=dict(hidden_sizes=[512, 256, 128], output_dim=6, t_dim=32, condition_dim=17,
                                            condition_hidden_dim=32, t_condition_hidden_dim=128)))))


def build_model(device):
    import generativerl_preview_amd as G
    from generativerl_preview_amd import _lib as L
    torch.manual_seed(0)
    m = G.DiffusionModel(qgpo_config(device))
    with torch.no_grad():  # keep sampled actions O(1) (last_block x 0.1)
        m.model.model["backbone"].last_block.weight.mul_(0.1)
        m.model.model["backbone"].last_block.bias.mul_(0.1)
    m = m.to(device)
    m.model.compute_dtype = L.BF16
    return m


def cpu_baseline(model, rows=32768, chunk=4096):
    """The CPU oracle (port of the reference's torch path) on a bounded sample of the same workload:
    `rows` rows x 100 Euler steps, chunked at 4096 like the reference's generate_fake_action."""
    from oracle import grl_oracle as O
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(threads)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = qgpo_config("cpu")
    g = torch.Generator().manual_seed(1)
    cond = torch.randn(rows, 17, generator=g)
    x0 = torch.randn(rows, 6, generator=g)
    t_span = torch.linspace(0.0, 1.0, N_SOLVER_STEPS + 1)
    t0 = time.time()
    with torch.no_grad():
        for i in range(0, rows, chunk):
            O.sample_forward_process(sd, cfg, "diffusion", t_span, x_0=x0[i:i + chunk], condition=cond[i:i + chunk])
    dt = time.time() - t0
    return dict(value=rows / dt, unit="samples/s", cores=threads, kind="port",
                sample="%d rows x %d Euler steps (chunks of %d), oracle/grl_oracle.py eager fp32 torch, %.1f s"
                       % (rows, N_SOLVER_STEPS, chunk, dt),
                action_steps_per_s=rows * N_SOLVER_STEPS / dt)


def pmc_traffic(rows):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (separate --pmc runs of this script at
    131072 states = 4194304 rows: scripts/profile_bench.sh), scaled linearly to this launch's rows: the counters
    cannot be read from inside the timed run.  2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE, KB -> bytes."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_bench_bf16_v3.json")
    try:
        with open(path) as f:
            prof = json.load(f)
        pm = next(v for k, v in prof.items() if k.startswith("pmc"))
        per_row = (2.0 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024.0 / 4194304.0
        return per_row * rows, "profiles/r01_bench_bf16_v3.json (separate PMC passes, scaled by rows)"
    except Exception:  # noqa: BLE001 -- the profile is optional
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--states", type=int, default=N_STATES, help="states per GPU (default: the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    device = "cuda:%d" % local
    torch.cuda.set_device(local)

    from generativerl_preview_amd import engine
    model = build_model(device)
    gen = torch.Generator(device=device).manual_seed(1 + rank)
    states = torch.randn(args.states, 17, device=device, generator=gen)  # this rank's block of states
    t_span = torch.linspace(0.0, 1.0, N_SOLVER_STEPS + 1)
    rows = args.states * N_CAND

    def step():
        # QGPO candidate generation: K = 32 actions per state, one fused launch over N*K rows
        return model.sample(t_span=t_span, batch_size=N_CAND, condition=states)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    engine.KERNEL_EVENTS = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    events, engine.KERNEL_EVENTS = engine.KERNEL_EVENTS, None
    assert out.shape == (N_CAND, args.states, 6) and bool(torch.isfinite(out).all())
    if dist is not None:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        kernel_ms = sum(a.elapsed_time(b) for a, b in events) / max(1, len(events))
        achieved = rows * N_SOLVER_STEPS * FLOP_PER_ACTION_STEP / (kernel_ms * 1e-3) / 1e12
        total_rows = rows * world * args.steps
        traffic, traffic_src = pmc_traffic(rows)
        res = {
            "metric": "action samples/sec (100-step ODE)",
            "value": total_rows / elapsed, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "action_steps_per_s": total_rows * N_SOLVER_STEPS / elapsed,
            "config": {"workload": "C2 QGPO HalfCheetah-shaped sampling: %d states x %d candidates per GPU, "
                                   "100-step Euler ODE on linspace(0,1,101), linear VP-SDE noise net [512,256,128], "
                                   "obs 17 / act 6, unguided, finals only" % (args.states, N_CAND),
                       "rows_per_gpu": rows, "solver_steps": N_SOLVER_STEPS, "parallelism": "batch-sharded x%d, no collective" % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_unit": "bytes per launch",
                         "traffic_source": traffic_src,
                         "kernel": "k_sample_bf16", "kernel_ms": kernel_ms,
                         "algorithmic_flop_per_launch": rows * N_SOLVER_STEPS * FLOP_PER_ACTION_STEP},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(model)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
