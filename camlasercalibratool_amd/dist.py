"""Multi-GPU layer: independent T_cl problems shard across the GPUs of a node (one process
per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for
tests).  There is no data-path collective: every rank solves its contiguous shard on its own
GPU, and the fixed-size result records are gathered once at the end.

Result record (12 doubles per problem):
    pose[7], final_cost, initial_cost, num_iterations, termination, global_problem_index
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np

RECORD = 12


def shard_problems(n_problems: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of problem indices owned by `rank` (sizes differ by <= 1)."""
    base, extra = divmod(n_problems, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def pack_records(poses: np.ndarray, summaries, lo: int) -> np.ndarray:
    p = len(summaries)
    out = np.zeros((p, RECORD))
    out[:, :7] = np.asarray(poses).reshape(p, 7)
    for k in range(p):
        s = summaries[k]
        out[k, 7:] = (s.final_cost, s.initial_cost, s.num_iterations, s.termination, lo + k)
    return out


def gather_result_records(local: np.ndarray, n_problems: int, device: Optional[str] = None) -> np.ndarray:
    """all_gather of the per-problem result records -> [n_problems, 12] on every rank, ordered
    by global problem index.  Shards may differ in size by one; they are padded to equal
    length for the collective."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(local, dtype=np.float64).reshape(-1, RECORD)
    world = dist.get_world_size()
    cap = (n_problems + world - 1) // world
    buf = torch.full((cap, RECORD), -1.0, dtype=torch.float64)
    loc = torch.as_tensor(np.asarray(local, dtype=np.float64).reshape(-1, RECORD))
    buf[: loc.shape[0]] = loc
    if device is not None:
        buf = buf.to(device)
    out = torch.empty((world * cap, RECORD), dtype=torch.float64, device=buf.device)
    dist.all_gather_into_tensor(out, buf)
    out = out.cpu().numpy()
    out = out[out[:, 11] >= 0]
    order = np.argsort(out[:, 11], kind="stable")
    out = out[order]
    assert out.shape[0] == n_problems, (out.shape, n_problems)
    return out


def solve_sharded(records: np.ndarray, offsets: np.ndarray, poses0: np.ndarray, options=None,
                  solve_fn: Optional[Callable] = None, device_index: Optional[int] = None) -> np.ndarray:
    """Solve P independent problems across all ranks.  `records`/`offsets`/`poses0` describe
    the FULL batch on every rank (each rank touches only its shard).  Returns the gathered
    [P,12] result records on every rank.

    solve_fn(records_shard, offsets_shard, poses_shard, options) -> (poses, summaries) defaults
    to the GPU batched solver; tests inject a stub to exercise the sharding on CPU/gloo."""
    import torch.distributed as dist

    P = len(offsets) - 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_problems(P, rank, world)
    off = np.ascontiguousarray(offsets[lo : hi + 1] - offsets[lo], dtype=np.int64)
    rec = np.ascontiguousarray(records[offsets[lo] : offsets[hi]])
    p0 = np.ascontiguousarray(np.asarray(poses0).reshape(P, 7)[lo:hi])
    device = None
    if solve_fn is None:
        from .solver import Solver

        dev = device_index if device_index is not None else rank

        def solve_fn(r, o, p, opt):  # noqa: E306
            with Solver(dev) as sv:
                sv.upload_batched(r, o)
                return sv.solve_batched(p, opt)

        device = f"cuda:{dev}"
    if hi > lo:
        poses, sms = solve_fn(rec, off, p0, options)
        local = pack_records(poses, sms, lo)
    else:
        local = np.zeros((0, RECORD))
    return gather_result_records(local, P, device)
