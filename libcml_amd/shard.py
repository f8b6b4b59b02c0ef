"""Sequence-shard fan-out: independent keyframe windows, one process per GPU, no data-path collective.
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests) is used only for the start/stop
barrier of the measurement and for gathering per-rank counts — there is no exchange step on this path to accelerate."""
import os
import time


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when launched directly."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def shards_for_rank(n_shards, rank, world):
    """Round-robin assignment of shard ids to ranks (weak scaling: bench.py uses one shard per rank)."""
    return [s for s in range(n_shards) if s % world == rank]


class Group:
    """Barrier + max/sum reductions; degenerates to no-ops for world == 1 so N=1 needs no process group."""

    def __init__(self, backend=None, device=None):
        self.rank, self.local_rank, self.world = env_world()
        self.dist = None
        self.device = device
        # CML_SHARD_FORCE_DIST=1: form the process group even for one rank (exercises the RCCL path on a single-GPU box)
        if self.world > 1 or os.environ.get("CML_SHARD_FORCE_DIST") == "1":
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if backend is None:
                # CML_SHARD_BACKEND=gloo: the CPU backend on a GPU box (tests: two ranks that SHARE one GPU — RCCL refuses two ranks on one device)
                backend = os.environ.get("CML_SHARD_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            if not dist.is_initialized():
                dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
            self.dist = dist
            self.torch = torch
            self.backend = backend

    def _tensor(self, v, dtype=None):
        t = self.torch
        dev = self.device if self.backend == "nccl" else "cpu"
        return t.tensor([v], dtype=dtype or t.float64, device=dev)

    def barrier(self):
        if self.dist is not None:
            # a 1-element all-reduce is the barrier (works for nccl and gloo alike).  The tensor is allocated ONCE: the closing barrier of a
            # timed region sits inside the region (the bench contract), and a fresh device tensor per call is a host-to-device copy on top
            # of the collective (tools/probe_barrier.py: 41 us per barrier with one rank, most of it not the all-reduce)
            if getattr(self, "_bar", None) is None:
                self._bar = self._tensor(0.0, dtype=self.torch.float32)
            self.dist.all_reduce(self._bar)
            if self.backend == "nccl":
                self.torch.cuda.synchronize()

    def max(self, v):
        if self.dist is None:
            return v
        x = self._tensor(float(v))
        self.dist.all_reduce(x, op=self.dist.ReduceOp.MAX)
        return float(x.item())

    def sum(self, v):
        if self.dist is None:
            return v
        x = self._tensor(float(v))
        self.dist.all_reduce(x, op=self.dist.ReduceOp.SUM)
        return float(x.item())

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()


def timed_region(group, sync, run_steps):
    """barrier + device sync on both sides, MAX over ranks of the elapsed wall time (the bench contract)."""
    sync()
    group.barrier()
    t0 = time.perf_counter()
    run_steps()
    sync()
    group.barrier()
    dt = time.perf_counter() - t0
    return group.max(dt)
