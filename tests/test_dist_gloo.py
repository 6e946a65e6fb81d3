"""Multi-process (world_size 2, gloo, CPU) test of the problem sharding + result gather used
for the 8-GPU batched configuration.  The per-shard solve is stubbed with the CPU oracle so
that the plumbing (shard boundaries, padding, ordering, collective) runs without a GPU."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_problems_partition():
    from camlasercalibratool_amd.dist import shard_problems

    for P in (0, 1, 7, 8, 65536, 1025):
        for W in (1, 2, 3, 8):
            spans = [shard_problems(P, r, W) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == P
            for a, b in zip(spans[:-1], spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, P, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    import oracle
    from camlasercalibratool_amd import dist as cdist, simdata as sd

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        loaded = []

        def load_shard(lo, hi):  # every rank generates ONLY its own problems (pure function of the global index)
            loaded.append((lo, hi))
            return sd.sim_shard_records(5, lo, hi, 8, 40, 0.01)[:3]

        def stub(r, o, p, opt):  # stands in for Solver.solve_batched on this rank's shard
            poses, sms = [], []
            for k in range(len(o) - 1):
                res = oracle.solve(r[o[k]:o[k + 1]], p[k], linear_solver="ne")
                poses.append(res.pose)
                sms.append(res.summary)
            return np.array(poses), sms

        out = cdist.solve_sharded(load_shard, P, None, solve_fn=stub)
        assert loaded == [cdist.shard_problems(P, rank, world)]
        if rank == 0:
            q.put(out)
        else:
            q.put(out[:, 11].copy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("P,world", [(5, 2), (6, 2), (13, 8)])
def test_sharded_solve_and_gather(P, world):
    """world 2, and 8 ranks with a problem count 8 does not divide (shards of 2 and 1 problems, padding in the gather)."""
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, P, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=400) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = next(o for o in outs if o.ndim == 2)
    assert full.shape == (P, 12) and np.array_equal(full[:, 11], np.arange(P))
    assert sum(o.ndim == 1 for o in outs) == world - 1
    for idx in (o for o in outs if o.ndim == 1):  # every other rank holds the same ordered result
        assert np.array_equal(idx, np.arange(P))
    # every problem solved exactly once, by the right rank, and equal to a single-process solve
    sys.path.insert(0, ROOT)
    import oracle
    from camlasercalibratool_amd import simdata as sd

    sh = sd.sim_shard(5, 0, P, 8, 40, 0.01)
    x0 = sh.start_poses()
    for k in range(P):
        rec = oracle.flatten(sh.problem(k), False, False)
        ref = oracle.solve(rec, x0[k], linear_solver="ne")
        assert np.array_equal(full[k, :7], ref.pose)
        assert full[k, 7] == ref.summary.final_cost and full[k, 9] == ref.summary.num_iterations
