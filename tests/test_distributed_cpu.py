"""world_size-2 gloo tests of the multi-GPU path's host logic (runs on CPU): row sharding and the
single flat gradient all-reduce."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from generativerl_preview_amd.distributed import allreduce_gradients, attach_flat_grads, shard_rows


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    data = torch.arange(40, dtype=torch.float32).reshape(8, 5) / 10.0
    lo, hi = shard_rows(8)
    loss = lin(data[lo:hi]).pow(2).sum(dim=1).mean()   # mean over the LOCAL rows
    loss.backward()
    allreduce_gradients(lin.parameters())
    if rank == 0:
        full = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
        full.load_state_dict(lin.state_dict())
        full(data).pow(2).sum(dim=1).mean().backward()  # the reference's mean over the GLOBAL batch
        err = max(float((a.grad - b.grad).abs().max()) for a, b in zip(lin.parameters(), full.parameters()))
        out.put(err)
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_matches_global_batch_mean():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err < 1e-5, err


@pytest.mark.parametrize("n,world", [(10, 4), (1048576, 8), (3, 8), (0, 2)])
def test_shard_rows_partitions_exactly(n, world):
    blocks = [shard_rows(n, r, world) for r in range(world)]
    assert blocks[0][0] == 0 and blocks[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
    sizes = [hi - lo for lo, hi in blocks]
    assert max(sizes) - min(sizes) <= 1


# ----------------------------------------------------------------------------- the package's own flat-gradient path
def _flat_worker(rank, world, port, out):
    try:
        _flat_worker_body(rank, world, port, out)
    except BaseException as e:  # noqa: BLE001 -- report instead of letting the parent wait for its timeout
        import traceback
        out.put(("error", traceback.format_exc()))
        raise


def _flat_worker_body(rank, world, port, out):
    """Each rank: the ICFM policy on CPU tensors, the oracle's loss over an UNEVEN local shard written into the
    model's FlatGrads buffer (what grlhip_loss_fwd_bwd does on the GPU), then ONE in-place all-reduce."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests"), os.path.join(root, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import golden_util as gu
    import util_models as um
    from oracle import grl_oracle as O
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    name = "small_icfm"
    kind, cfg, m = um.build(name)
    fg = attach_flat_grads(m.model)
    assert all(p.grad is not None and p.grad.data_ptr() >= fg.flat.data_ptr() for p in m.model.parameters() if p.requires_grad)
    B = 21
    gen = torch.Generator().manual_seed(3)
    x0, x1, cond = torch.randn(B, 6, generator=gen), torch.rand(B, 6, generator=gen), torch.randn(B, 17, generator=gen)
    t, eps = torch.rand(B, generator=gen), torch.randn(B, 6, generator=gen)
    lo, hi = (0, 8) if rank == 0 else (8, B)      # 8 and 13 rows: ranks hold different row counts

    def oracle_grads(sl):
        sd = {k: v.clone().requires_grad_(not k.endswith("t_encoder.W")) for k, v in gu.weights(name).items()}
        loss = O.icfm_flow_matching_loss(lambda tt, y: O.intrinsic_forward(sd, tt, y, cond[sl]), x0[sl], x1[sl], t[sl],
                                         eps[sl], cfg["path"]["sigma"])
        keys = [k for k in sd if sd[k].requires_grad]
        return loss.detach(), dict(zip(keys, torch.autograd.grad(loss, [sd[k] for k in keys])))

    loss, grads = oracle_grads(slice(lo, hi))
    named = {"model.model." + k: p for k, p in m.model.model.named_parameters()}
    with torch.no_grad():
        for k, g in grads.items():
            named[k].grad.copy_(g)                 # lands in the flat buffer: .grad is a view of it
    before = fg.flat.data_ptr()
    gloss = fg.allreduce(hi - lo, loss)
    assert fg.flat.data_ptr() == before
    if rank == 0:
        full_loss, full = oracle_grads(slice(0, B))
        err = max(float((named[k].grad - g).abs().max() / (g.abs().max() + 1e-12)) for k, g in full.items())
        out.put((err, abs(float(gloss) - float(full_loss))))
    # zero_grad(set_to_none=True) drops the views; ensure() re-installs them on a zeroed buffer
    torch.optim.SGD(m.model.parameters(), lr=0.1).zero_grad(set_to_none=True)
    fg.ensure()
    assert float(fg.flat[:fg.n].abs().max()) == 0.0
    assert all(p.grad is not None for p in m.model.parameters() if p.requires_grad)
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_buffer_allreduce_equals_global_batch_gradient():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flat_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err, loss_err = q.get(timeout=180)
    assert err != "error", loss_err
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err < 1e-5 and loss_err < 1e-6, (err, loss_err)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (how the driver starts it) must spawn the
    two ranks itself, rendezvous, time with barriers + MAX over ranks and print ONE JSON line: rehearsed with --dry
    (gloo on CPU, stubbed step)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry", "--states", "16",
                        "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["warmup"] == 1 and res["dry"] is True
    assert res["ms_per_step"] >= 10.0          # MAX over ranks: rank 1's stub sleeps 10 ms per step, rank 0's 5 ms
    assert abs(res["value"] - 16 * 32 * 2 * 3 / (res["ms_per_step"] * 3e-3)) < 1e-6 * res["value"]
    # what the collective backend itself observed, and the C3 training leg: every rank runs the step and its ONE
    # all-reduce of the flat gradient buffer, weighted by rows (ranks hold 8 and 13 rows, stub gradients 1 and 2)
    assert res["backend"] == "gloo" and res["rccl_world_size"] == 2
    c3 = [e for e in res["secondary"] if e["name"] == "C3_icfm_train"]
    assert len(c3) == 1 and c3[0]["n_gpus"] == 2
    assert c3[0]["allreduce"] == {"world_size": 2, "backend": "gloo", "elements": c3[0]["allreduce"]["elements"],
                                  "collectives_per_step": 1}
    chk = c3[0]["allreduce_check"]
    assert abs(chk["expected"] - (8 * 1.0 + 13 * 2.0) / 21) < 1e-12
    assert abs(chk["got_grad"] - chk["expected"]) < 1e-6 and abs(chk["got_loss"] - chk["expected"]) < 1e-6
    # a mismatching launcher world size is refused, not silently run
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry", "--states", "16"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0


def test_flat_gradient_buffer_clip_equals_torch_clip_grad_norm():
    """FlatGrads.clip_grad_norm_ (one norm over the flat buffer every .grad aliases) against
    torch.nn.utils.clip_grad_norm_ over the same parameters: same total norm, same clipped gradients, with a norm above
    and below the threshold."""
    import copy
    import util_models as um
    from generativerl_preview_amd.distributed import attach_flat_grads
    kind, cfg, m = um.build("small_icfm")
    fg = attach_flat_grads(m.model, "cpu")
    gen = torch.Generator().manual_seed(2)
    for scale, max_norm in ((1.0, 1.0), (1e-4, 1.0), (3.0, 0.25)):
        fg.zero()
        fg.ensure()
        for p in m.model.parameters():
            if p.requires_grad:
                p.grad.copy_(torch.randn(p.shape, generator=gen) * scale)
        twin = [p.grad.detach().clone().requires_grad_(False) for p in m.model.parameters() if p.requires_grad]
        params = [torch.nn.Parameter(torch.zeros_like(g)) for g in twin]
        for q, g in zip(params, twin):
            q.grad = g.clone()
        want = torch.nn.utils.clip_grad_norm_(params, max_norm)
        got = fg.clip_grad_norm_(max_norm)
        assert abs(float(got) - float(want)) <= 1e-6 * float(want)
        for q, p in zip(params, [p for p in m.model.parameters() if p.requires_grad]):
            assert torch.allclose(p.grad, q.grad, rtol=1e-6, atol=1e-12)
