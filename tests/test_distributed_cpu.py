"""world_size-2 gloo tests (CPU, no GPU): the only cross-rank exchanges of the path -- bench.py's max-over-ranks
timing and the all-reduce of the packed normal-equation accumulators (rosdyn_amd/gram.py)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from rosdyn_amd.gram import allreduce_normal_equations
    # sharding: contiguous, complete, balanced
    sizes = bench.shard_sizes(1000003, world)
    assert sum(sizes) == 1000003 and max(sizes) - min(sizes) <= 1
    # time = slowest rank
    t = bench.max_over_ranks(1.0 + rank, dist, torch.device("cpu"))
    # accumulators: every rank holds the Gram of its own rows; the all-reduce must give the Gram of all rows
    P, rows = 7, 50
    rng = np.random.default_rng(1234)
    A = rng.standard_normal((world * rows, P))
    b = rng.standard_normal(world * rows)
    Ar, br = A[rank * rows:(rank + 1) * rows], b[rank * rows:(rank + 1) * rows]
    G, c, bb, cnt = allreduce_normal_equations(torch.from_numpy(Ar.T @ Ar), torch.from_numpy(Ar.T @ br),
                                               torch.tensor([br @ br]), rows, dist)
    ok = (np.allclose(G.numpy(), A.T @ A, rtol=1e-13, atol=1e-13) and np.allclose(c.numpy(), A.T @ b, rtol=1e-13, atol=1e-13)
          and abs(float(bb.item()) - b @ b) < 1e-12 and cnt == world * rows and t == float(world))
    # the packed payload reduced IN PLACE (what bench.py's config4 does every step: no packing kernels, no host read)
    from rosdyn_amd.gram import allreduce_packed, packed_buffer, unpack_normal_equations
    buf = packed_buffer(P, rows, torch.device("cpu"))
    for _ in range(2):                       # the buffer is re-used: the count slot is refilled, G / c / bb are overwritten
        buf[:P * P] = torch.from_numpy(Ar.T @ Ar).reshape(-1)
        buf[P * P:P * P + P] = torch.from_numpy(Ar.T @ br)
        buf[P * P + P] = float(br @ br)
        assert allreduce_packed(buf, dist, count=rows) is buf
        G2, c2, bb2, cnt2 = unpack_normal_equations(buf, P)
        ok = ok and torch.equal(G2, G) and torch.equal(c2, c) and cnt2 == world * rows
    # the R-factor exchange (the TSQR alternative): every rank the factor of ITS rows, one all-gather, the fold of the stack
    # in rank order -- against the factor of all rows, and the same bits on both ranks
    from rosdyn_amd.gram import allgather_fold_r_factors
    Mr = np.column_stack([Ar, br])
    Rr = np.linalg.qr(Mr, mode="r")
    Rall = allgather_fold_r_factors(torch.from_numpy(Rr), dist).numpy()
    Mall = np.column_stack([A, b])
    ok = ok and np.allclose(np.tril(Rall, -1), 0.0) and np.abs(Rall.T @ Rall - Mall.T @ Mall).max() <= 1e-12 * np.abs(Mall.T @ Mall).max()
    both = [torch.empty_like(torch.from_numpy(Rall)) for _ in range(world)]
    dist.all_gather(both, torch.from_numpy(Rall))
    ok = ok and all(torch.equal(both[0], x) for x in both)
    q.put((rank, ok))
    dist.destroy_process_group()


def test_gloo_world2_allreduce_and_timing():
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]
