"""rdyn_regressor_tsqr launched directly vs replayed from a captured HIP graph (14 launches per call, seven of which leave at once):
what the launch gaps are worth.  us per call, N = 1e6 / 66 000 / 16 000 samples, 6 joints."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain
from rosdyn_amd._lib import lib

chain = Chain(os.path.join(ROOT, "tests/fixtures/ur10_like.urdf"), "base_link", "wrist_3_link", (0, 0, -9.806))
n = 6
for N in (1000000, 66000, 16000):
    q, dq, ddq, tau = (torch.rand((N, n), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(4))
    ws = torch.empty((lib().rdyn_regressor_tsqr_workspace_bytes(chain._h),), dtype=torch.uint8, device="cuda")
    out = torch.empty((61, 61), dtype=torch.float64, device="cuda")
    chain.getRegressorTsqr(q, dq, ddq, tau, workspace=ws)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            R = chain.getRegressorTsqr(q, dq, ddq, tau, workspace=ws)
    res = {}
    for rep in range(3):
        for name, fn in (("direct", lambda: chain.getRegressorTsqr(q, dq, ddq, tau, workspace=ws)), ("graph", g.replay)):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"N = {N:8d}   direct {['%.1f' % x for x in res['direct']]}   graph {['%.1f' % x for x in res['graph']]}", flush=True)
