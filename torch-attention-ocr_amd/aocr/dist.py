"""Data parallelism over the image batch: the ONE exchange step of the hot path.

The reference is single-GPU (src/train.lua:246).  Every sample is independent except the BatchNorm batch
statistics and the scalar loss, so each rank runs the full step on its slice with d(loss) scaled by
1/global_batch (the reference divides by the step's batch size, model.lua:645-647) and the ranks sum their flat
gradient vectors with one all-reduce placed between feval and the per-group clip (optim_sgd.lua:38 -> :40), so
every rank clips and updates identically.  `torch.distributed` backend "nccl" is RCCL over xGMI on ROCm; the
same code runs on gloo for the CPU tests.  BatchNorm statistics stay per-rank (documented deviation).
"""
from __future__ import annotations

import torch


def world_size() -> int:
    d = torch.distributed
    return d.get_world_size() if (d.is_available() and d.is_initialized()) else 1


def grad_scale(local_batch: int) -> float:
    """d(loss)/d(pred) scale so that the SUM over ranks equals the reference's 1/batch_size of the global batch."""
    return 1.0 / (local_batch * world_size())


def exchange(flat_grads: torch.Tensor, loss: torch.Tensor) -> None:
    """In-place sum over ranks of the flat gradient vector (one bucket: 12.18 M fp32 = 48.7 MB at He=256) and the loss."""
    if world_size() > 1:
        torch.distributed.all_reduce(flat_grads)
        torch.distributed.all_reduce(loss)
