"""Throughput probe of the batched local IK (not a bench line): UR10-like 6-DOF, N poses, seeds 0.25 rad around the goal."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rosdyn_amd import Chain
from rosdyn_amd.samples import uniform_pm1

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
chain = Chain("tests/fixtures/ur10_like.urdf", "base_link", "tool0")
q_goal = torch.from_numpy(uniform_pm1(1, (N, 6))).cuda()
seeds = q_goal + 0.25 * torch.from_numpy(uniform_pm1(2, (N, 6))).cuda()
T = chain.getTransformation(q_goal)
for cap in (1, 2, 4, 8, 30):
    chain.computeLocalIk(T, seeds, toll=1e-6, max_iterations=cap)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        sol, st, it = chain.computeLocalIk(T, seeds, toll=1e-6, max_iterations=cap)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("cap %2d: %.3f ms  %.3g poses/s  converged %.4f  mean QP updates %.2f  failures %d" % (
        cap, dt * 1e3, N / dt, (st == 1).double().mean().item(), it.double().mean().item(), (st < 0).sum().item()))
