#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): action samples/sec for the 100-step Euler ODE of the QGPO
HalfCheetah-shaped policy (obs 17, act 6, TemporalSpatialResidualNet [512,256,128], linear VP-SDE,
noise head), 1 048 576 synthetic states x 32 candidate actions PER GPU, through
``DiffusionModel.sample`` -> C ABI -> the bf16 MFMA persistent integrator.

    python bench.py --gpus N --steps K --warmup W

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ...` (RANK / WORLD_SIZE in the
environment) or started plainly, in which case this process spawns the N rank processes itself before touching the
GPU (generativerl_preview_amd/launch.py).  One "step" = one full pass (all rows, all 100 solver steps).  Prints ONE
JSON line on rank 0.  Sampling shards by state blocks; there is no data-path collective (scaling: weak).
`--dry` (tests): gloo on CPU, the HIP step replaced by a stub -- exercises spawn, rendezvous, barriers, the
MAX-over-ranks timing and the JSON line without a GPU.
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


TRAINED_WEIGHTS = os.path.join(ROOT, "tests", "golden", "weights_qgpo_noise.npz")


def build_model(device):
    """The C2 policy with the TRAINED weights the parity tests are asserted on (tests/golden/weights_qgpo_noise.npz:
    the qgpo_noise golden model, a few hundred Adam steps of the reference's own loss, 100-step samples end at
    |x| = O(1)); the file also holds the energy-guidance net, which this unguided config does not use."""
    import numpy as np
    import generativerl_preview_amd as G
    from generativerl_preview_amd import _lib as L
    m = G.DiffusionModel(qgpo_config(device))
    z = np.load(TRAINED_WEIGHTS)
    sd = {k: torch.from_numpy(np.ascontiguousarray(z[k], dtype=np.float32)) for k in z.files
          if not k.startswith("energy_guidance.")}
    m.load_state_dict(sd)     # strict: every key of the reference's state_dict layout must be there
    m = m.to(device)
    m.model.compute_dtype = L.BF16
    return m


def check_against_oracle(model, states, out, seed, n_check=64, tol=2e-2):
    """64 rows of the TIMED output against the CPU oracle (outside the timed region).  The last timed step drew its
    x_T right after torch.manual_seed(seed): the same draw is repeated here.  Row r of the launch has x_T[r], the
    condition states[r // K] (repeat_interleave) and lands in out.reshape(-1, D)[r] (diffusion_model.py:205-234, 369-416)."""
    from oracle import grl_oracle as O
    rows = states.shape[0] * N_CAND
    torch.manual_seed(seed)
    x_T = model.gaussian_generator(rows)
    idx = torch.arange(n_check, device=states.device, dtype=torch.int64) * ((rows - 1) // max(1, n_check - 1))  # first .. last row
    assert int(idx.max()) < rows
    x0, cond = x_T[idx].cpu(), states[idx // N_CAND].cpu()
    got = out.reshape(-1, out.shape[-1])[idx].cpu()
    del x_T
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    want = O.sample_forward_process(sd, qgpo_config("cpu"), "diffusion", torch.linspace(0.0, 1.0, N_SOLVER_STEPS + 1),
                                    x_0=x0, condition=cond)[-1]
    err = float((got - want).abs().max())
    assert err <= tol, "timed output differs from the oracle: max-abs %.3e > %.1e" % (err, tol)
    return {"rows_checked": n_check, "max_abs_err": err, "tolerance": tol, "max_abs_action": float(want.abs().max()),
            "against": "oracle/grl_oracle.py (fp32 torch restatement of the reference), same x_T / states / grid"}


def cpu_info():
    model, phys = "unknown", set()
    try:
        pid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                phys.add((pid, line.split(":", 1)[1].strip()))
    except OSError:
        pass
    return model, len(phys) or None, os.cpu_count()


def cpu_threads():
    return min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))


def cpu_baseline(model, rows=65536, chunk=4096):
    """The CPU oracle (port of the reference's torch path) on a bounded sample of the same workload:
    `rows` rows x 100 Euler steps, chunked at 4096 like the reference's generate_fake_action."""
    from oracle import grl_oracle as O
    threads = cpu_threads()
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
    cpu, phys, logical = cpu_info()
    return dict(value=rows / dt, unit="samples/s", cores=threads, kind="port",
                sample="%d rows x %d Euler steps (chunks of %d), oracle/grl_oracle.py eager fp32 torch, %.1f s"
                       % (rows, N_SOLVER_STEPS, chunk, dt),
                cpu_model=cpu, physical_cores=phys, logical_cores=logical, threads=threads,
                action_steps_per_s=rows * N_SOLVER_STEPS / dt)


def f32_roofline(tf):
    """fp32 batch engine: fp32-equivalent TFLOP/s.  Its products are six bf16 MFMAs each (three-way split),
    so the matrix-core ceiling is the dense bf16 peak / 6; the fp32-MFMA peak (157.3) is kept beside it for reference.  The
    measured bound of these kernels is the per-CU operand fetch, not either MFMA peak (profiles/r02_gemm_x6_microbench.md)."""
    peak = PEAK_BF16_TFLOPS / 6.0
    return {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s (fp32-equivalent)", "frac": tf / peak,
            "peak_note": "dense bf16 MFMA peak / 6 MFMAs per fp32 product", "frac_of_fp32_mfma_peak_157.3": tf / 157.3}


PMC_PROFILE = "profiles/r03_bench_pmc.json"


def pmc_traffic(rows):
    """HBM-side bytes per launch of k_sample_bf16 from the committed rocprofv3 PMC passes (separate --pmc runs of this
    script, counters summed over the k_sample_bf16 dispatches only: scripts/profile_bench.sh), scaled linearly to this
    launch's rows: the counters cannot be read from inside the timed run.  2 x FETCH_SIZE (gfx950 correction) +
    WRITE_SIZE, KB -> bytes."""
    path = os.path.join(ROOT, PMC_PROFILE)
    try:
        with open(path) as f:
            pm = json.load(f)["pmc_k_sample_bf16"]
        per_row = (2.0 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024.0 / pm["rows"]
        return per_row * rows, PMC_PROFILE + " (separate PMC passes filtered to k_sample_bf16, scaled by rows)"
    except Exception:  # noqa: BLE001 -- the profile is optional
        return None, None


def _time_gpu(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


def secondary_results(device, model):
    """Other BASELINE configs / engines on the same box, one object each (never the headline): the fp32 engine on the
    C2 net (the precision trade) and C5 (GVP [1024] D=64, 100-step ODE, batch 262144).  Small enough to add ~10 s to
    the run.  C3 (the training step) is c3_training(): it runs at every world size."""
    import generativerl_preview_amd as G
    from generativerl_preview_amd import _lib as L
    res = []
    t_span = torch.linspace(0.0, 1.0, N_SOLVER_STEPS + 1)
    gen = torch.Generator(device=device).manual_seed(7)
    try:   # fp32 batch engine on the C2 net
        rows = 262144
        x0 = torch.randn(rows, 6, device=device, generator=gen)
        cond = torch.randn(rows, 17, device=device, generator=gen)
        model.model.compute_dtype = L.F32
        dt, _ = _time_gpu(lambda: model.sample(t_span=t_span, x_0=x0, condition=cond), 1)
        model.model.compute_dtype = L.BF16
        tf = rows * N_SOLVER_STEPS * FLOP_PER_ACTION_STEP / dt / 1e12
        res.append({"name": "fp32_engine", "workload": "C2 net, %d rows x 100-step Euler ODE on the fp32 batch engine" % rows,
                    "dtype": "f32", "value": rows / dt, "unit": "samples/s", "action_steps_per_s": rows * N_SOLVER_STEPS / dt,
                    "roofline": f32_roofline(tf)})
    except Exception as e:  # noqa: BLE001
        res.append({"name": "fp32_engine", "error": repr(e)})
    try:   # C5: GVP velocity net [1024], D = 64
        t_enc = dict(type="GaussianFourierProjectionTimeEncoder", args=dict(embed_dim=32, scale=30.0))
        cfg = dict(device=device, x_size=64, alpha=1.0, path=dict(type="gvp"),
                   solver=dict(type="ODESolver", args=dict(library="torchdyn")),
                   model=dict(type="velocity_function", args=dict(t_encoder=t_enc, backbone=dict(
                       type="TemporalSpatialResidualNet", args=dict(hidden_sizes=[1024], output_dim=64, t_dim=32)))))
        torch.manual_seed(0)
        m5 = G.DiffusionModel(cfg).to(device)
        m5.model.compute_dtype = L.BF16
        rows = 262144
        x0 = torch.randn(rows, 64, device=device, generator=gen)
        dt, _ = _time_gpu(lambda: m5.sample(t_span=t_span, x_0=x0), 2)
        tf = rows * N_SOLVER_STEPS * 6950912 / dt / 1e12
        res.append({"name": "C5_gvp_wide", "workload": "C5 GVP velocity net [1024], D=64, batch %d, 100-step Euler ODE" % rows,
                    "dtype": "bf16", "value": rows / dt, "unit": "samples/s", "action_steps_per_s": rows * N_SOLVER_STEPS / dt,
                    "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                 "frac": tf / PEAK_BF16_TFLOPS}})
        del m5, x0
    except Exception as e:  # noqa: BLE001
        res.append({"name": "C5_gvp_wide", "error": repr(e)})
    return res


def c3_training(ranks, dry=False, steps=20, warmup=5):
    """BASELINE config C3 on every rank of this launch: GMPO ICFM flow_matching_loss fwd + bwd into ONE flat fp32
    gradient buffer, ONE in-place all-reduce of that buffer (RCCL through torch.distributed at world > 1, weighted by
    rows; nothing at world 1), Adam.  65 536 rows per GPU (weak scaling), barrier + synchronize on both sides, MAX over
    ranks.  Returns the `secondary` entry on rank 0.  dry: gloo on CPU, the fused loss replaced by a stub that writes
    rank-dependent gradients into the same FlatGrads buffer, so the all-reduce and its row weighting are exercised."""
    import generativerl_preview_amd as G
    from generativerl_preview_amd.distributed import attach_flat_grads
    rank, world, device = ranks.rank, ranks.world, ranks.device
    hidden = [64, 32, 16] if dry else [512, 256, 128]
    model_cfg = qgpo_config(device)["model"]
    model_cfg = dict(model_cfg, type="velocity_function")
    model_cfg["args"]["backbone"]["args"]["hidden_sizes"] = hidden
    cfg = dict(device=device, x_size=6, alpha=1.0, path=dict(sigma=0.1),
               solver=dict(type="ODESolver", args=dict(library="torchdyn")), model=model_cfg)
    torch.manual_seed(0)    # identical initial weights on every rank
    m3 = G.IndependentConditionalFlowModel(cfg).to(device)
    fg = attach_flat_grads(m3.model)   # every .grad is a view of ONE flat buffer: one zero, one all-reduce
    if dry:
        opt = torch.optim.Adam(m3.parameters(), lr=1e-4)
    else:                              # Adam over ONE flat parameter buffer: one launch (grlhip_adam_step)
        from generativerl_preview_amd.optim import FlatAdam
        opt = FlatAdam(m3.model, lr=1e-4)
    B = (8 + 5 * rank) if dry else 65536
    gen = torch.Generator(device=device).manual_seed(11 + rank)
    x1 = torch.rand(B, 6, device=device, generator=gen) * 2 - 1
    cond = torch.randn(B, 17, device=device, generator=gen)
    check = {}

    def train_step():
        fg.zero()
        if dry:
            fg.grads.fill_(float(rank + 1))
            loss = torch.tensor(float(rank + 1))
        else:
            loss = m3.flow_matching_loss(torch.randn(B, 6, device=device, generator=gen), x1, cond)
            loss.backward()                  # one fused pass into the flat buffer
        gloss = fg.allreduce(B, loss)        # ONE in-place all-reduce (returns the local loss at world 1)
        if dry:
            check["grad"], check["loss"] = float(fg.grads[0]), float(gloss)
        opt.step()
        return gloss

    elapsed, loss = ranks.timed(train_step, warmup, steps)
    assert bool(torch.isfinite(loss))
    dt2 = None
    if not dry:   # the opt-in two-piece backward, reported beside the default and never instead of it
        m3.model.backward_split = 2
        elapsed2, loss2 = ranks.timed(train_step, 2, steps)
        m3.model.backward_split = 3
        assert bool(torch.isfinite(loss2))
        dt2 = elapsed2 / steps
    if rank != 0:
        return None
    dt = elapsed / steps
    entry = {"name": "C3_icfm_train",
             "workload": "C3 GMPO ICFM flow_matching_loss fwd+bwd + flat-gradient all-reduce + Adam, batch %d per GPU, %d GPU(s)" % (B, world),
             "dtype": "f32", "value": B * world / dt, "unit": "rows/s", "ms_per_step": dt * 1e3, "n_gpus": world,
             "scaling": "weak", "final_loss": float(loss.detach()),
             "allreduce": {"world_size": world, "backend": ranks.backend, "elements": int(fg.flat.numel()),
                           "collectives_per_step": 1 if world > 1 else 0}}
    if dry:
        rows = [8 + 5 * r for r in range(world)]
        entry["allreduce_check"] = {"got_grad": check["grad"], "got_loss": check["loss"],
                                    "expected": sum(n * (r + 1.0) for r, n in enumerate(rows)) / sum(rows)}
        return entry
    entry["roofline"] = f32_roofline(B * 9759936 / dt / 1e12)   # per GPU, whole step over wall time
    entry["backward_split"] = 3
    entry["two_piece_backward"] = {
        "ms_per_step": dt2 * 1e3, "value": B * world / dt2, "unit": "rows/s",
        "note": "the same step with grlhip_plan_set_backward_split(plan, 2): two bf16 pieces per fp32 value (three MFMAs "
                "per product) in the backward GEMMs only; loss unchanged, gradients ~1e-5 relative (opt-in, not the default)"}
    if world == 1:
        try:   # the same step on the host cores through the oracle (bounded sample)
            from oracle import grl_oracle as O
            torch.set_num_threads(cpu_threads())
            sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point and not k.endswith("t_encoder.W"))
                  for k, v in m3.state_dict().items()}
            Bc = 8192
            g = torch.Generator().manual_seed(3)
            cx0, cx1, cc = torch.randn(Bc, 6, generator=g), torch.rand(Bc, 6, generator=g) * 2 - 1, torch.randn(Bc, 17, generator=g)
            ct, ce = torch.rand(Bc, generator=g), torch.randn(Bc, 6, generator=g)
            leaves = [v for v in sd.values() if v.requires_grad]
            copt = torch.optim.Adam(leaves, lr=1e-4)
            t0 = time.time()
            n = 0
            while time.time() - t0 < 6.0 or n < 2:
                copt.zero_grad()
                O.icfm_flow_matching_loss(lambda tt, y: O.intrinsic_forward(sd, tt, y, cc), cx0, cx1, ct, ce, 0.1).backward()
                copt.step()
                n += 1
            cdt = (time.time() - t0) / n
            entry["cpu_baseline"] = {"value": Bc / cdt, "unit": "rows/s", "cores": cpu_threads(), "kind": "port",
                                     "sample": "%d steps of batch %d, oracle loss + torch autograd + Adam, %.1f s" % (n, Bc, cdt * n)}
        except Exception as e:  # noqa: BLE001
            entry["cpu_baseline"] = {"error": repr(e)}
    return entry


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--states", type=int, default=N_STATES, help="states per GPU (default: the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--dry", action="store_true", help="CPU rehearsal of the launch path (gloo, stubbed step)")
    args = ap.parse_args()

    from generativerl_preview_amd import launch
    if launch.needs_self_launch(args.gpus):   # nothing has touched the GPU yet
        sys.exit(launch.self_launch(args.gpus))
    ranks = launch.Ranks(args.gpus, dry=args.dry)
    rank, world, device = ranks.rank, ranks.world, ranks.device
    rows = args.states * N_CAND
    t_span = torch.linspace(0.0, 1.0, N_SOLVER_STEPS + 1)

    if args.dry:
        engine = model = None
        w = torch.randn(64, 64)

        def step():  # stand-in for the HIP launch: a little CPU work, rank-dependent so that MAX differs from local
            time.sleep(0.005 * (1 + rank))
            return (torch.randn(N_CAND, args.states, 6) @ torch.eye(6)) + 0.0 * w.sum()
    else:
        from generativerl_preview_amd import engine
        model = build_model(device)
        gen = torch.Generator(device=device).manual_seed(1 + rank)
        states = torch.randn(args.states, 17, device=device, generator=gen)  # this rank's block of states

        def step():
            # QGPO candidate generation: K = 32 actions per state, one fused launch over N*K rows
            return model.sample(t_span=t_span, batch_size=N_CAND, condition=states)

    for _ in range(args.warmup):
        out = step()
    ranks.barrier()
    events = []
    if engine is not None:
        # HIP events on the launch stream right around the library call that launches k_sample_bf16 (the roofline's
        # kernel time): a wrapper installed here, for the timed steps only -- the product module carries no bench hooks
        plan_sample = engine.NetPlan.sample

        def timed_sample(self, packed, x0, *a, **k):
            st = torch.cuda.current_stream(x0.device)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(st)
            r = plan_sample(self, packed, x0, *a, **k)
            ev[1].record(st)
            events.append(ev)
            return r
        engine.NetPlan.sample = timed_sample
    check_seed = 4242 + rank
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == args.steps - 1:
            torch.manual_seed(check_seed)   # the last step's x_T draw is repeated by check_against_oracle()
        out = step()
    ranks.barrier()
    elapsed = ranks.max_over_ranks(time.perf_counter() - t0)
    if engine is not None:
        engine.NetPlan.sample = plan_sample
    assert out.shape == (N_CAND, args.states, 6) and bool(torch.isfinite(out).all())

    parity = None
    if engine is not None and rank == 0:
        parity = check_against_oracle(model, states, out, check_seed)   # asserts 2e-2; outside the timed region
    del out
    # C3 on every rank (the one gradient all-reduce per step is the only collective of the whole path)
    c3 = None if args.no_secondary else c3_training(ranks, dry=args.dry, steps=3 if args.dry else 20,
                                                    warmup=1 if args.dry else 5)

    if rank == 0:
        total_rows = rows * world * args.steps
        res = {
            "metric": "action samples/sec (100-step ODE)",
            "value": total_rows / elapsed, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "action_steps_per_s": total_rows * N_SOLVER_STEPS / elapsed,
            "backend": ranks.backend, "rccl_world_size": ranks.collective_world,
            "config": {"workload": "C2 QGPO HalfCheetah-shaped sampling: %d states x %d candidates per GPU, "
                                   "100-step Euler ODE on linspace(0,1,101), linear VP-SDE noise net [512,256,128], "
                                   "obs 17 / act 6, unguided, finals only" % (args.states, N_CAND),
                       "weights": "trained (tests/golden/weights_qgpo_noise.npz), synthetic states",
                       "rows_per_gpu": rows, "solver_steps": N_SOLVER_STEPS, "parallelism": "batch-sharded x%d, no collective" % world},
        }
        if args.dry:
            res["dry"] = True
            res["roofline"] = None
            if c3 is not None:
                res["secondary"] = [c3]
        else:
            kernel_ms = sum(a.elapsed_time(b) for a, b in events) / max(1, len(events))
            achieved = rows * N_SOLVER_STEPS * FLOP_PER_ACTION_STEP / (kernel_ms * 1e-3) / 1e12
            traffic, traffic_src = pmc_traffic(rows)
            res["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_unit": "bytes per launch",
                               "traffic_source": traffic_src,
                               "kernel": "k_sample_bf16", "kernel_ms": kernel_ms,
                               "algorithmic_flop_per_launch": rows * N_SOLVER_STEPS * FLOP_PER_ACTION_STEP}
            res["parity"] = parity
            if world == 1 and not args.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(model)
            sec = []
            if world == 1 and not args.no_secondary:
                sec = secondary_results(device, model)
            if c3 is not None:
                sec.append(c3)
            if sec:
                res["secondary"] = sec
        print(json.dumps(res), flush=True)
    ranks.close()


if __name__ == "__main__":
    main()
