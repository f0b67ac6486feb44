"""Rank plumbing for bench.py / multi-rank runs: one process per GPU, torch.distributed
(backend "nccl" = RCCL on ROCm; "gloo" on CPU for tests).  The data path has no collective yet
(every rank owns an independent subdomain-sized problem); what crosses ranks is timing only."""
from __future__ import annotations

import os


class Ranks:
    def __init__(self, backend=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        self.backend = backend
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            kw = {}
            if backend == "nccl" and device is not None:
                kw["device_id"] = device
            dist.init_process_group(backend or "gloo", rank=self.rank, world_size=self.world, **kw)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        if self.device is not None and self.device.type == "cuda":
            self.torch.cuda.synchronize()

    def max_over_ranks(self, x: float) -> float:
        if self.world == 1:
            return float(x)
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.device if self.backend == "nccl" else None)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x: float) -> float:
        if self.world == 1:
            return float(x)
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.device if self.backend == "nccl" else None)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def aggregate_throughput(units_per_rank_step: float, steps: int, seconds_max: float, world: int) -> float:
    """whole-job units/s: units all ranks processed / max-over-ranks time (weak scaling)."""
    return world * units_per_rank_step * steps / seconds_max
