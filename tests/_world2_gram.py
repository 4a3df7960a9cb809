"""Helper of tests/test_gpu_configs.py::test_config4_share_two_ranks_on_one_gpu (run under torch.distributed.run, 2 ranks, ONE GPU):
every rank computes the fused regressor -> Gram of its shard on device 0 through the C-ABI, the packed normal equations are summed by
one all-reduce, rank 0 compares with the Gram of the whole batch computed by one call and writes the verdict as JSON."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain                                         # noqa: E402
from rosdyn_amd.gram import allreduce_normal_equations, solve_base_parameters   # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    torch.cuda.set_device(0)
    chain = Chain(os.path.join(ROOT, "tests", "fixtures", "ur10_like.urdf"), "base_link", "wrist_3_link", (0.0, 0.0, -9.806))
    n, P = chain.getActiveJointsNumber(), 10 * chain.getJointsNumber()
    n_total = 200000 + 37                                              # ragged: the shards differ in size
    gen = torch.Generator(device="cuda").manual_seed(4242)            # same stream on both ranks -> same full batch
    q, dq, ddq = (torch.rand((n_total, n), dtype=torch.float64, device="cuda", generator=gen) * 2 - 1 for _ in range(3))
    tau = chain.getJointTorque(q, dq, ddq)                             # noise-free "measurements"
    base, rem = divmod(n_total, world)
    sizes = [base + (1 if r < rem else 0) for r in range(world)]
    lo = sum(sizes[:rank])
    sl = slice(lo, lo + sizes[rank])
    G, c, bb = chain.getRegressorGram(q[sl].contiguous(), dq[sl].contiguous(), ddq[sl].contiguous(), tau[sl].contiguous())
    torch.cuda.synchronize()
    Gs, cs, bbs, count = allreduce_normal_equations(G.cpu(), c.cpu(), bb.cpu(), sizes[rank], dist)   # ONE all-reduce (gloo, CPU tensors)
    if rank == 0:
        G1, c1, bb1 = chain.getRegressorGram(q, dq, ddq, tau)
        torch.cuda.synchronize()
        G1, c1, bb1 = G1.cpu(), c1.cpu(), bb1.cpu()
        x, rank_G = solve_base_parameters(Gs, cs)
        pi = chain.getNominalParameters()
        Gn = Gs.numpy()
        out = {"world": world, "n_total": n_total, "count": count,
               "rel_G": float((Gs - G1).norm() / G1.norm()), "rel_c": float((cs - c1).norm() / c1.norm()),
               "rel_bb": float(abs(bbs[0] - bb1[0]) / abs(bb1[0])),
               # identified parameters reproduce the normal equations' right-hand side: G x = c (x is the minimum-norm solution)
               "param_err": float(np.abs(Gn @ x - cs.numpy()).max() / np.abs(cs.numpy()).max()), "rank_G": rank_G,
               "nominal_residual": float(np.abs(Gn @ pi - cs.numpy()).max() / np.abs(cs.numpy()).max())}
        with open(sys.argv[1], "w") as f:
            json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
