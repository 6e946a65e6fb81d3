"""Multi-GPU layer (BASELINE.json configs[3]): independent T_cl problems shard across the GPUs of a
node, one process per GPU.  There is no collective on the data path: every rank LOADS ONLY ITS OWN
contiguous shard, keeps it resident in its GPU's HBM and solves it with clc_solve_batched; the
fixed-size result records are gathered once per solve over xGMI.

The gather is a C-ABI entry point (clc_gather_results, include/clc.h): ncclAllGather on the solver's own
stream, straight from the device buffer the batched solver leaves its results in.  torch.distributed is
used for what it is here for — process rendezvous (rank / world size, distributing the 128-byte RCCL
unique id).  With a CPU process group (gloo; tests, dry runs) the same records are gathered with
torch.distributed.all_gather_into_tensor instead.

Result record (12 doubles per problem, clc_result_record):
    pose[7], final_cost, initial_cost, num_iterations, termination, global_problem_index
"""
from __future__ import annotations

import os

from typing import Callable, Optional, Tuple

import numpy as np

RECORD = 12


def shard_problems(n_problems: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of problem indices owned by `rank` (sizes differ by <= 1)."""
    base, extra = divmod(n_problems, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def shard_capacity(n_problems: int, world: int) -> int:
    """Records every rank contributes to the all-gather (the largest shard; smaller ones are padded)."""
    return max(1, (n_problems + world - 1) // world)


def pack_records(poses: np.ndarray, summaries, lo: int) -> np.ndarray:
    p = len(summaries)
    out = np.zeros((p, RECORD))
    out[:, :7] = np.asarray(poses).reshape(p, 7)
    for k in range(p):
        s = summaries[k]
        out[k, 7:] = (s.final_cost, s.initial_cost, s.num_iterations, s.termination, lo + k)
    return out


def order_records(gathered: np.ndarray, n_problems: int) -> np.ndarray:
    """Rank-major gathered records (padding has global index -1) -> [n_problems, 12] by global index."""
    out = np.asarray(gathered, dtype=np.float64).reshape(-1, RECORD)
    if out.shape[0] == n_problems and np.array_equal(out[:, 11], np.arange(n_problems)):
        return out  # equal contiguous shards in rank order: already the global order, no padding
    out = out[out[:, 11] >= 0]
    out = out[np.argsort(out[:, 11], kind="stable")]
    if out.shape[0] != n_problems or not np.array_equal(out[:, 11], np.arange(n_problems)):
        raise RuntimeError(f"gather returned {out.shape[0]} records for {n_problems} problems (or indices not 0..P-1)")
    return out


def _rank_world() -> Tuple[int, int]:
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return 0, 1


def gather_result_records(local: np.ndarray, n_problems: int, device: Optional[str] = None, ordered: bool = True) -> np.ndarray:
    """torch.distributed path (any backend; what the CPU/gloo tests run): all_gather of the per-problem
    result records -> [n_problems, 12] on every rank, ordered by global problem index (ordered=False: the raw
    rank-major [world * cap, 12] buffer with its padding records, global index -1, like clc_gather_results)."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return order_records(local, n_problems) if ordered else np.asarray(local, dtype=np.float64).reshape(-1, RECORD)
    world = dist.get_world_size()
    cap = shard_capacity(n_problems, world)
    buf = torch.full((cap, RECORD), -1.0, dtype=torch.float64)
    loc = torch.as_tensor(np.asarray(local, dtype=np.float64).reshape(-1, RECORD))
    buf[: loc.shape[0]] = loc
    if device is not None:
        buf = buf.to(device)
    out = torch.empty((world * cap, RECORD), dtype=torch.float64, device=buf.device)
    dist.all_gather_into_tensor(out, buf)
    return order_records(out.cpu().numpy(), n_problems) if ordered else out.cpu().numpy()


def exchange_unique_id(rank: int, world: int) -> bytes:
    """Rank 0 draws the RCCL unique id (clc_comm_unique_id); the 128 bytes travel over the process group's store."""
    from .solver import comm_unique_id

    if world == 1:
        return comm_unique_id()
    import torch.distributed as dist

    box = [comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


class ShardSolver:
    """One rank's share of a sharded batch of `n_problems` independent problems: a Solver on this rank's GPU with
    the shard resident, plus the RCCL communicator for the result gather.  Construct once, solve many times."""

    def __init__(self, n_problems: int, device_index: Optional[int] = None, rank: Optional[int] = None,
                 world: Optional[int] = None, use_rccl: bool = True, root: Optional[int] = None):
        """use_rccl=False (dry runs on a CPU process group, e.g. two ranks sharing one GPU, which RCCL refuses):
        the records are gathered with torch.distributed instead of clc_gather_results.
        root: None = every rank ends up with every record; r = only rank r does (clc_comm_set_root: ncclGather to r, one host copy at r):
        the solve calls then return None on the other ranks."""
        from .solver import Comm, Solver

        r, w = _rank_world()
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        self.n_problems = n_problems
        self.lo, self.hi = shard_problems(n_problems, self.rank, self.world)
        self.cap = shard_capacity(n_problems, self.world)
        if device_index is None:
            # the HIP device is the LOCAL rank (torchrun exports LOCAL_RANK); the global rank only names the device on one
            # node with every GPU visible — the fallback for launchers that set neither
            device_index = int(os.environ["LOCAL_RANK"]) if "LOCAL_RANK" in os.environ else (0 if self.world == 1 else self.rank)
        self.device_index = device_index
        self.solver = Solver(device_index)
        self.comm = Comm(self.solver, exchange_unique_id(self.rank, self.world), self.rank, self.world) if use_rccl else None
        self.root = root
        if self.comm is not None and root is not None:
            self.comm.set_root(root)

    def _deliver(self, out, ordered: bool):
        """What a solve call returns on this rank: the gathered records (ordered by global index or raw), or None where a root is set
        and this rank is not it."""
        if out is None or (self.root is not None and self.rank != self.root):
            return None
        return order_records(out, self.n_problems) if ordered else out

    def upload(self, records: np.ndarray, offsets: np.ndarray):
        """This rank's shard only: records of problems [lo, hi), offsets relative to the shard."""
        assert len(offsets) - 1 == self.hi - self.lo, "upload() takes exactly this rank's shard"
        self.solver.upload_batched(records, offsets)

    def solve(self, poses0: np.ndarray, options=None, ordered: bool = True, copy: bool = True, inplace: bool = False) -> np.ndarray:
        """clc_solve_batched on the shard + clc_gather_results: [n_problems, 12] on every rank (ordered=False:
        the raw rank-major [world * cap, 12] buffer, padding records with global index -1 included).
        inplace=True: the start poses are written into the handle's own pinned buffer and solved there
        (clc_batched_host_buffers): last_poses / last_summaries are then views that the next solve overwrites."""
        self.last_poses, self.last_summaries = None, None
        if self.hi > self.lo:
            if inplace:
                pb, _ = self.solver.batched_buffers()
                pb[:] = np.asarray(poses0, dtype=np.float64).reshape(pb.shape)
                self.last_poses, self.last_summaries = self.solver.solve_batched_inplace(options)
            else:
                self.last_poses, self.last_summaries = self.solver.solve_batched(poses0, options)
        if self.comm is None:
            local = pack_records(self.last_poses, self.last_summaries, self.lo) if self.hi > self.lo else np.zeros((0, RECORD))
            return gather_result_records(local, self.n_problems, ordered=ordered)
        out = self.comm.gather_results(self.lo, self.cap, copy=copy)
        return self._deliver(out, ordered)

    def solve_gather(self, poses0: np.ndarray, options=None, ordered: bool = True, copy: bool = True) -> np.ndarray:
        """The same step as solve() through ONE call, clc_solve_batched_gather (include/clc.h): the kernel's epilogue writes the result
        records into the gather buffer, all-gather in place, one copy to the host — no per-problem poses / summaries cross PCIe.
        -> the records as solve(); the local shard's totals (evaluation passes, iterations, not converged) in `last_stats`.
        Needs the RCCL communicator (use_rccl=True)."""
        assert self.comm is not None, "solve_gather() needs the RCCL communicator; dry runs on a CPU process group use solve()"
        self.last_poses, self.last_summaries = None, None
        out, self.last_stats = self.comm.solve_gather(poses0 if self.hi > self.lo else None, self.lo, self.cap, options, copy=copy)
        return self._deliver(out, ordered)

    def solve_gather_pipelined(self, poses0: np.ndarray, options=None, ordered: bool = False):
        """A stream of steps (clc_solve_batched_gather_pipelined): enqueues this step and returns the PREVIOUS step's records (None on the
        first call, and on ranks other than a set root) — the copy of the other ranks' records to the host overlaps this step's kernel.
        `last_stats` are the previous step's totals.  flush() returns the last step's records."""
        assert self.comm is not None, "solve_gather_pipelined() needs the RCCL communicator"
        out, st = self.comm.solve_gather_pipelined(poses0 if self.hi > self.lo else None, self.lo, self.cap, options)
        if st is not None:
            self.last_stats = st
        return self._deliver(out, ordered)

    def flush(self, ordered: bool = False):
        out, st = self.comm.flush(self.cap)
        if st is not None:
            self.last_stats = st
        return self._deliver(out, ordered)

    def close(self):
        if self.comm is not None:
            self.comm.close()
        self.solver.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def solve_sharded(load_shard: Callable[[int, int], Tuple[np.ndarray, np.ndarray, np.ndarray]], n_problems: int,
                  options=None, solve_fn: Optional[Callable] = None, device_index: Optional[int] = None) -> np.ndarray:
    """Solve `n_problems` independent problems across all ranks, one shot.

    load_shard(lo, hi) -> (records [N,8], offsets [hi-lo+1] relative to the shard, poses0 [hi-lo,7]) is called
    with THIS rank's block only: no rank ever materialises the full batch (C4 is 42 GB of records).
    Returns the gathered [n_problems, 12] result records on every rank.

    solve_fn(records, offsets, poses0, options) -> (poses, summaries), if given, replaces the GPU solver and the
    records are gathered with torch.distributed (how the CPU/gloo tests exercise the sharding); otherwise the shard is
    solved by clc_solve_batched on this rank's GPU and gathered by clc_gather_results (RCCL)."""
    rank, world = _rank_world()
    lo, hi = shard_problems(n_problems, rank, world)
    if solve_fn is not None:
        local = np.zeros((0, RECORD))
        if hi > lo:
            rec, off, p0 = load_shard(lo, hi)
            poses, sms = solve_fn(rec, off, p0, options)
            local = pack_records(poses, sms, lo)
        return gather_result_records(local, n_problems)
    with ShardSolver(n_problems, device_index, rank, world) as ss:
        p0 = np.zeros((0, 7))
        if hi > lo:
            rec, off, p0 = load_shard(lo, hi)
            ss.upload(rec, off)
        return ss.solve(p0, options)
