#!/bin/bash
mkdir -p gpurun_out/r3
timeout 1200 python -m pytest tests/test_gpu_tsqr.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r3/run19_tests.txt
K=tools/_build/kbench
L=rosdyn_amd/variants/librdyn_probes.so
{
timeout 300 $K ident 1 rosdyn_amd/librdyn_hip.so
} > gpurun_out/r3/run19_kbench.txt 2>&1
python - > gpurun_out/r3/run19_ident.txt 2>&1 <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from rosdyn_amd import Chain
from rosdyn_amd.components import ComponentSet
c = Chain("tests/fixtures/ur10_like.urdf", "base_link", "wrist_3_link", (0, 0, -9.806))
N, n = 1000000, 6
q, dq, ddq, tau = (torch.rand((N, n), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(4))
comps = ComponentSet([dict(type=0, joint=j, min_velocity=1e-3, max_velocity=10.0, parameters=[0.1, 0.2]) for j in range(n)], n)
for name, f in (("identification R factor [Y | 6 friction | tau]", lambda: c.getIdentificationTsqr(comps, q, dq, ddq, tau)), ("regressor R factor", lambda: c.getRegressorTsqr(q, dq, ddq, tau))):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); print(name, (time.perf_counter() - t0) / 5 * 1e3, "ms")
PY
