#!/usr/bin/env python3
"""GPU probe: the Gram paths on the same box -- fused persistent kernel vs the two-kernel chunked path."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain          # noqa: E402
from rosdyn_amd._lib import lib       # noqa: E402
from tools.probe import timeit        # noqa: E402

for name, urdf, base, tool, N in (("cfg2", "ur10_like.urdf", "base_link", "wrist_3_link", 1000000),
                                  ("cfg3", "panda_like.urdf", "link0", "link7", 4000000)):
    chain = Chain(os.path.join(ROOT, "tests/fixtures", urdf), base, tool, (0, 0, -9.806))
    n = chain.getActiveJointsNumber()
    q, dq, ddq, tm = (torch.rand((n, N), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(4))
    ref = None
    # (a third variant -- the two kernels software-pipelined over two images and a helper stream -- measured 1.50 ms vs
    #  1.41 ms serial vs 1.12 ms fused at n = 6, N = 1e6 and was removed)
    for label, chunk, env in (("lds tile (default)", 0, {}), ("global image", 0, {"RDYN_GRAM_PATH": "image"}),
                              ("two kernels chunk=262144", 262144, {})):
        for k, v in env.items():
            os.environ[k] = v
        ws = torch.empty((lib().rdyn_regressor_gram_workspace_bytes(chain._h, chunk),), dtype=torch.uint8, device="cuda")
        out = chain.getRegressorGram(q, dq, ddq, tm, layout="element", chunk_samples=chunk, workspace=ws)
        t = timeit(lambda: chain.getRegressorGram(q, dq, ddq, tm, layout="element", chunk_samples=chunk, out=out, workspace=ws), reps=5, warm=2)
        if ref is None:
            ref = out[0].clone()
        print("%s %-26s %8.1f us -> %.3e evals/s   rel diff vs fused %.1e" % (name, label, t * 1e6, N / t, float((out[0] - ref).norm() / ref.norm())))
        for k in env:
            del os.environ[k]
        del ws
