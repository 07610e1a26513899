"""Multi-GPU plumbing for the ``full_gp`` path: one process per GPU (``torch.distributed``; backend
"nccl" = RCCL on ROCm, "gloo" in CPU tests).

The reference's multi-GPU story is task parallelism over independent GPs - one ``mp.Process`` per GPU
over battery systems (``gp_runner.py:246-298``), 9 independent GPs per system
(``src/batt_models/battgp_full.py:41-60``) - so the data path needs NO collective: cells are dealt to
ranks and only timings / small result vectors are reduced or gathered.
"""

from __future__ import annotations

import os
from typing import Sequence

import numpy as np


def env_rank_world() -> tuple[int, int, int]:
    return (
        int(os.environ.get("RANK", "0")),
        int(os.environ.get("WORLD_SIZE", "1")),
        int(os.environ.get("LOCAL_RANK", "0")),
    )


def cells_for_rank(cell_ids: Sequence[int], rank: int, world: int) -> list[int]:
    """Round-robin deal of cells (pack = -1 first, then 1..n) to ranks: with 9 GPs on 8 GPUs rank 0
    takes the pack model and cell 8."""
    return [c for i, c in enumerate(cell_ids) if i % world == rank]


def init(backend: str | None = None, device=None, force: bool = False):
    """Initialise the default process group when launched by torch.distributed.run; returns the
    ``torch.distributed`` module or None for a single process.  ``force`` (or ``BGP_FORCE_GROUP=1``) builds the
    group even for one process, so that every collective of the sharded path goes through the backend
    (a 1-GPU box can put RCCL under the real call sequence that way)."""
    rank, world, _ = env_rank_world()
    if world <= 1 and not (force or os.environ.get("BGP_FORCE_GROUP") == "1"):
        return None
    import torch.distributed as dist

    if not dist.is_initialized():
        if backend is None:
            import torch

            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl" and device is not None:
            kwargs["device_id"] = device
        if world <= 1:  # forced single-process group: the launcher's rendezvous variables may be missing
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            kwargs.update(rank=0, world_size=1)
        dist.init_process_group(backend, **kwargs)
    return dist


def barrier(dist) -> None:
    if dist is not None:
        dist.barrier()


def max_over_ranks(dist, value: float, device="cpu") -> float:
    """Slowest rank's time - the job time of a weak-scaling run."""
    if dist is None:
        return float(value)
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_vectors(dist, vec: np.ndarray, device="cpu") -> list[np.ndarray]:
    """All ranks' (equal-length) result vectors on every rank - e.g. the 2 x 300 floats of
    ``predict_r0_op`` per cell; the only data that ever crosses ranks on this path."""
    if dist is None:
        return [np.asarray(vec)]
    import torch

    t = torch.as_tensor(np.ascontiguousarray(vec), dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.cpu().numpy() for o in out]
