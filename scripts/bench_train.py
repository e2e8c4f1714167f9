#!/usr/bin/env python
"""Secondary benchmark, BASELINE config C3: GMPO ICFM flow-model training step -- flow_matching_loss
forward + backward through the fp32 batch engine, ONE flat gradient all-reduce over RCCL (N > 1), Adam --
batch 65 536 per GPU (weak scaling).  Same contract as bench.py:

    python scripts/bench_train.py --gpus N --steps K --warmup W     (N > 1: spawns its ranks, or torch.distributed.run)

One JSON line on rank 0: training rows/s over all ranks, and the fp32-MFMA roofline of the step.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

FLOP_PER_ROW = 9759936        # fwd + bwd = 3 x 2 x 1 626 656 MAC, SURVEY.md section 8d
PEAK_F32_TFLOPS = 2500.0 / 6   # fp32 products = six bf16 MFMAs each: dense bf16 peak / 6; fp32-MFMA peak: 157.3
BATCH = 65536


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help="rows per GPU")
    ap.add_argument("--backward-split", type=int, default=3, choices=(2, 3),
                    help="bf16 pieces per fp32 value in the backward GEMMs (3 = fp32 rounding level, the default)")
    ap.add_argument("--graph", action="store_true",
                    help="capture the whole step once with torch.cuda.graph and time its replays (one GPU only)")
    args = ap.parse_args()
    from generativerl_preview_amd import launch
    if launch.needs_self_launch(args.gpus):   # nothing has touched the GPU yet
        sys.exit(launch.self_launch(args.gpus))
    ranks = launch.Ranks(args.gpus)
    rank, world, device, dist = ranks.rank, ranks.world, ranks.device, ranks.dist

    import golden_lib as gl
    import generativerl_preview_amd as G
    from generativerl_preview_amd.distributed import attach_flat_grads
    torch.manual_seed(0)  # identical initial weights on every rank
    m = G.IndependentConditionalFlowModel(dict(gl.icfm_cfg(6, [512, 256, 128], 17), device=device)).to(device)
    m.model.backward_split = args.backward_split
    fg = attach_flat_grads(m.model)   # .grad of every parameter is a view of ONE flat buffer
    from generativerl_preview_amd.optim import FlatAdam
    use_graph = args.graph and world == 1
    opt = FlatAdam(m.model, lr=1e-4, capturable=use_graph)   # ONE launch over the flat parameter buffer (grlhip_adam_step)
    B = args.batch
    gen = torch.Generator(device=device).manual_seed(1 + rank)
    x1 = torch.rand(B, 6, device=device, generator=gen) * 2 - 1
    cond = torch.randn(B, 17, device=device, generator=gen)

    def step():
        fg.zero()
        x0 = torch.randn(B, 6, device=device) if use_graph else torch.randn(B, 6, device=device, generator=gen)
        loss = m.flow_matching_loss(x0, x1, cond)
        loss.backward()                      # one fused pass into the flat buffer
        gl_loss = fg.allreduce(B, loss)      # ONE in-place all-reduce (no-op at world 1)
        opt.step()
        return loss if gl_loss is None else gl_loss

    if use_graph:   # every launch of the step goes to the capture stream; the optimizer's step count lives on the device
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream(device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_loss = step()

        def replay():
            graph.replay()
            return static_loss
        elapsed, loss = ranks.timed(replay, args.warmup, args.steps)
    else:
        elapsed, loss = ranks.timed(step, args.warmup, args.steps)
    import ctypes
    from generativerl_preview_amd import _lib as L
    launches = 0
    for prog in (2, 3):   # training forward / backward programs of the policy net
        n_ops, n_l = ctypes.c_int32(), ctypes.c_int32()
        L.check(L.lib().grlhip_plan_program_info(m.model.plan.handle, prog, ctypes.byref(n_ops), ctypes.byref(n_l)))
        launches += n_l.value
    assert bool(torch.isfinite(loss))
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        achieved = B * FLOP_PER_ROW / (ms * 1e-3) / 1e12   # per GPU, whole step (incl. repack, all-reduce, Adam)
        print(json.dumps({
            "metric": "training rows/sec (ICFM flow_matching_loss fwd+bwd+all-reduce+Adam)",
            "value": B * world * args.steps / elapsed, "unit": "rows/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "backward_split": args.backward_split, "hip_graph_replay": bool(use_graph),
            "config": {"workload": "C3 GMPO ICFM training step, batch %d per GPU, net [512,256,128], obs 17 / act 6, sigma 0.1" % B,
                       "parallelism": "data parallel x%d, one flat fp32 all-reduce per step" % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_TFLOPS, "frac_of_fp32_mfma_peak_157.3": achieved / 157.3, "traffic": None,
                         "note": "whole step per GPU over wall time; per-kernel split: scripts/profile_train.sh"},
            "library_launches_per_step": launches + 4 + 4,
            "launch_note": "op programs %d (GEMM / dW + ordered reduce / elementwise) + inputs, loss, Fourier gradient (2) + "
                           "repack of the 49 packed linears after Adam (3 batched launches + the Fourier row); torch adds the "
                           "loss prep, Adam (multi-tensor) and the flat-buffer zero" % launches,
            "final_loss": float(loss.detach())}), flush=True)
    ranks.close()


if __name__ == "__main__":
    main()
