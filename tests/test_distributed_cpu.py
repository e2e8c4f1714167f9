"""world_size-2 gloo tests of the multi-GPU path's host logic (runs on CPU): row sharding and the
single flat gradient all-reduce."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from generativerl_preview_amd.distributed import allreduce_gradients, shard_rows


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
