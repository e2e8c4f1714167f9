"""One process per GPU: self-launch and rendezvous helpers shared by bench.py and scripts/bench_train.py.

The reference is single-process; this is new.  Two ways in:
  * `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* come from the environment;
  * `python bench.py --gpus N` with no WORLD_SIZE in the environment: ``self_launch`` starts N fresh child processes
    of the same command line with those variables set (rendezvous on 127.0.0.1) BEFORE anything in the parent has
    touched the GPU -- children are spawned with subprocess, never by replacing the running process -- waits for
    them and returns the worst exit code.  Rank 0's stdout is the parent's stdout, so its one JSON line is the
    command's one JSON line.
"""
import os
import socket
import subprocess
import sys
import time


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def needs_self_launch(gpus: int) -> bool:
    return gpus > 1 and "WORLD_SIZE" not in os.environ


def self_launch(gpus: int, argv=None, timeout=None) -> int:
    """Start `gpus` rank processes of this command line and wait for them.  Returns the exit code to exit with."""
    argv = list(sys.argv if argv is None else argv)
    port = _free_port()
    procs = []
    for rank in range(gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(gpus),
                   LOCAL_WORLD_SIZE=str(gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # the host driver of these machines supports dmabuf IPC only: without this RCCL's intra-node buffer exchange
        # fails with `hipIpcGetMemHandle: invalid argument`.  The image exports it already; kept for bare environments.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # ranks other than 0 keep stderr (diagnostics) but must not add lines to stdout
        procs.append(subprocess.Popen([sys.executable] + argv, env=env,
                                      stdout=None if rank == 0 else subprocess.DEVNULL))
    deadline = None if timeout is None else time.time() + timeout
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                r = p.poll()
                if r is None:
                    continue
                pending.remove(p)
                if r != 0:
                    rc = rc or r
                    for q in pending:      # one rank died: the others would hang in a collective
                        q.terminate()
            if deadline is not None and time.time() > deadline:
                rc = rc or 124
                for q in pending:
                    q.terminate()
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
    return rc


class Ranks:
    """The process group of this launch (world 1: no group, barrier = device synchronize only)."""

    def __init__(self, gpus: int, dry: bool = False):
        import torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.dry = dry
        self.dist = None
        self.backend = None          # "nccl" (= RCCL on ROCm) / "gloo" once a process group exists
        self.collective_world = 1    # world size the collective backend itself reports
        if self.world != gpus:
            raise SystemExit("--gpus %d does not match WORLD_SIZE=%d of the launcher" % (gpus, self.world))
        if dry:
            self.device = "cpu"
        else:
            self.device = "cuda:%d" % self.local
            torch.cuda.set_device(self.local)
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if dry:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                        device_id=torch.device("cuda", self.local))
            self.dist = dist
            self.world = dist.get_world_size()   # what the backend observed
            self.collective_world = self.world
            self.backend = dist.get_backend()

    def sync(self):
        import torch
        if not self.dry:
            torch.cuda.synchronize()

    def barrier(self):
        self.sync()
        if self.dist is not None:
            self.dist.barrier()
        self.sync()

    def max_over_ranks(self, seconds: float) -> float:
        import torch
        if self.dist is None:
            return seconds
        t = torch.tensor([seconds], device=self.device, dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def timed(self, step, warmup: int, steps: int):
        """`warmup` untimed steps, then exactly `steps` steps bracketed by barrier + synchronize on both sides;
        returns (elapsed seconds, MAX over ranks; last step's return value)."""
        out = None
        for _ in range(warmup):
            out = step()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0), out

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
