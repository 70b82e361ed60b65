"""Batch-axis sharding of ``log_prob`` over the GPUs of one node.

Rows are independent on this path (every reduction in the reference is over the feature axis:
coupling.py:95, affine.py:109,171, dist/normal.py:37), so rank r simply owns the contiguous row block
``[r*N/G, (r+1)*N/G)``; weights are replicated.  The only exchange is ONE all-reduce of one fp64 — the
summed log-likelihood, accumulated in fp64 on the device — over RCCL (backend "nccl" on ROCm).
Per-sample outputs stay sharded.  One process per GPU; ``torch.distributed`` is plumbing only.
"""
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_rows(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous row block of `rank`: sizes differ by at most one row, concatenation = [0, n_rows)."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class PendingSum:
    """Handle of an enqueued (local sum -> all-reduce); ``wait()`` returns the global sum tensor."""

    def __init__(self, out: torch.Tensor, work):
        self.out, self.work = out, work

    def wait(self) -> torch.Tensor:
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self.out


class ShardedLogProb:
    """sum_n log p(y_n) over a batch sharded across the ranks of `group`.

    `local_sum(y_local, out)` must add the fp64 sum of the local rows' log-probs into `out` (a 1-element
    fp64 tensor on y_local's device).  By default that is `flow.log_prob_sum` (fused HIP kernel)."""

    def __init__(self, flow=None, group: Optional[dist.ProcessGroup] = None,
                 local_sum: Optional[Callable[[torch.Tensor, torch.Tensor], None]] = None):
        if local_sum is None:
            if flow is None:
                raise ValueError('give a flow or a local_sum callable')
            local_sum = lambda y, out: flow.log_prob_sum(y, out)
        self.local_sum = local_sum
        self.group = group

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    @property
    def rank(self) -> int:
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    def my_rows(self, n_rows: int) -> Tuple[int, int]:
        return shard_rows(n_rows, self.rank, self.world)

    def log_prob_sum_async(self, y_local: torch.Tensor, out: torch.Tensor) -> 'PendingSum':
        """Enqueue the local sum and its all-reduce without making the compute stream wait for the collective: the
        8-byte exchange of batch i (latency-bound, ~tens of microseconds) then runs under the kernel of batch i+1.
        ``out`` must not be reused before ``.wait()`` of the handle that owns it."""
        out.zero_()
        self.local_sum(y_local, out)
        work = None
        if dist.is_initialized() and self.world > 1:
            work = dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return PendingSum(out, work)

    def log_prob_sum(self, y_local: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """All ranks return the global sum (fp64, 1 element)."""
        if out is None:
            out = torch.zeros(1, dtype=torch.float64, device=y_local.device)
        else:
            out.zero_()
        self.local_sum(y_local, out)
        if dist.is_initialized() and self.world > 1:
            dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
        return out
