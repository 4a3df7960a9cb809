"""Helper of tests/test_gpu_configs.py::test_config3_share_two_ranks_on_one_gpu (run under torch.distributed.run, 2 ranks, ONE GPU):
every rank computes the robust R factor of ITS shard of [Y | friction columns | tau] through the C-ABI on device 0
(rdyn_identification_tsqr: the preconditioned CholeskyQR route), the factors are exchanged by ONE all-gather (gloo on host copies: two
ranks cannot share one device under RCCL) and folded in rank order on every rank (rosdyn_amd.gram.allgather_fold_r_factors -- what
bench.py's config3_sharded leg does over RCCL); rank 0 compares with the factor of the whole batch computed by one call and with the
normal equations, every rank reports a checksum of its folded factor so that the test can see that both hold the same bits."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain                                         # noqa: E402
from rosdyn_amd.components import ComponentSet                       # noqa: E402
from rosdyn_amd.gram import allgather_fold_r_factors                 # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    torch.cuda.set_device(0)
    chain = Chain(os.path.join(ROOT, "tests", "fixtures", "panda_like.urdf"), "link0", "hand", (0.0, 0.0, -9.806))
    n, P = chain.getActiveJointsNumber(), 10 * chain.getJointsNumber()
    comps = ComponentSet([dict(type=0, joint=j, min_velocity=1e-3, max_velocity=5.0, parameters=[0.4, 0.9]) for j in range(n)], n)
    C = P + comps.columns
    n_total = 120000 + 13                                              # ragged: the shards differ in size
    gen = torch.Generator(device="cuda").manual_seed(777)             # same stream on both ranks -> same full batch
    q, dq, ddq, tau = (torch.rand((n_total, n), dtype=torch.float64, device="cuda", generator=gen) * 2 - 1 for _ in range(4))
    base, rem = divmod(n_total, world)
    sizes = [base + (1 if r < rem else 0) for r in range(world)]
    lo = sum(sizes[:rank])
    sl = slice(lo, lo + sizes[rank])
    R = chain.getIdentificationTsqr(comps, *(x[sl].contiguous() for x in (q, dq, ddq, tau)))
    torch.cuda.synchronize()
    Rall = allgather_fold_r_factors(R.cpu(), dist)                     # ONE all-gather + the fold of the stack, on every rank
    sums = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(sums, torch.tensor([float(Rall.sum()), float(Rall.abs().sum())], dtype=torch.float64))
    if rank == 0:
        R1 = chain.getIdentificationTsqr(comps, q, dq, ddq, tau).cpu()
        G, c, bb = (t.cpu() for t in chain.getIdentificationGram(comps, q, dq, ddq, tau))
        full = torch.zeros((C + 1, C + 1), dtype=torch.float64)
        full[:C, :C], full[:C, C], full[C, :C], full[C, C] = G, c, c, bb[0]
        s1, s2 = np.linalg.svd(R1.numpy(), compute_uv=False), np.linalg.svd(Rall.numpy(), compute_uv=False)
        keep = s1 > 1e-9 * s1[0]
        out = {"world": world, "n_total": n_total, "n1": C + 1,
               "upper": bool(torch.equal(torch.tril(Rall, -1), torch.zeros_like(Rall))),
               "rel_gram": float((Rall.t() @ Rall - full).abs().max() / full.abs().max()),
               "rel_one_call": float((Rall.t() @ Rall - R1.t() @ R1).abs().max() / full.abs().max()),
               "sv_err": float(np.abs(s2[keep] / s1[keep] - 1.0).max()),
               "same_bits_on_all_ranks": bool(all(torch.equal(sums[0], s) for s in sums))}
        with open(sys.argv[1], "w") as f:
            json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
