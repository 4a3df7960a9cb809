"""README quick start: the Python mirror of rosdyn::Chain on device tensors (run from the repository root on a GPU box)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rosdyn_amd import Chain
chain = Chain(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/fixtures/ur10_public.urdf"), "base_link", "tool0", (0, 0, -9.806))   # 9 chain joints, 6 input joints, P = 90
q, Dq, DDq = (torch.rand((1_000_000, 6), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(3))
Y, tau = chain.getRegressor(q, Dq, DDq, with_torque=True)        # (N, 90, 6): the reference's column-major 6 x 90 image per sample
G, c, bb = chain.getRegressorGram(q, Dq, DDq, tau)               # normal equations on the fp64 matrix cores, Y never stored
R1 = chain.getRegressorTsqr(q, Dq, DDq, tau)                     # R factor of [A | tau] without the normal equations
print(Y.shape, tau.shape, G.shape, R1.shape, float((R1.t() @ R1)[:90, :90].sub(G).abs().max() / G.abs().max()))
