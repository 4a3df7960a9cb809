#!/usr/bin/env python3
"""GPU probe: Gram path timings (materialised Y + rdyn_gram, and the chunked fused rdyn_regressor_gram)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain          # noqa: E402
from rosdyn_amd.gram import gram      # noqa: E402
from tools.probe import timeit        # noqa: E402


def flops_gram(n, P):
    return n * P * (P + 1) + 2 * n * P      # dense-syrk convention + A^T b


def run(name, urdf, base, tool, N, chunks):
    dev = torch.device("cuda:0")
    chain = Chain(os.path.join(ROOT, "tests/fixtures", urdf), base, tool, (0, 0, -9.806))
    n, P = chain.getActiveJointsNumber(), 10 * chain.getJointsNumber()
    q, dq, ddq, tm = (torch.rand((n, N), dtype=torch.float64, device=dev) * 2 - 1 for _ in range(4))
    print("== %s: n=%d P=%d N=%d" % (name, n, P, N))
    if N * n * P * 8 < 40e9:
        Y = torch.empty((P, n, N), dtype=torch.float64, device=dev)
        tau = torch.empty((n, N), dtype=torch.float64, device=dev)
        t1 = timeit(lambda: chain.getRegressor(q, dq, ddq, layout="element", out=Y, tau_out=tau), reps=5, warm=2)
        A, b = Y.reshape(P, n * N), tm.reshape(n * N)
        out = gram(A, b)
        ws = torch.empty((1 << 26,), dtype=torch.uint8, device=dev)
        t2 = timeit(lambda: gram(A, b, out=out, workspace=ws), reps=5, warm=2)
        print("  materialise Y %.1f us + gram %.1f us (%.2f TFLOP/s, reads %.0f GB/s) = %.1f us -> %.3e evals/s" % (
            t1 * 1e6, t2 * 1e6, flops_gram(n, P) * N / t2 / 1e12, n * N * (P + 1) * 8 / t2 / 1e9, (t1 + t2) * 1e6, N / (t1 + t2)))
        del Y, A
    for ch in chunks:
        nbytes = 1 << 20
        from rosdyn_amd._lib import lib
        ws = torch.empty((lib().rdyn_regressor_gram_workspace_bytes(chain._h, ch),), dtype=torch.uint8, device=dev)
        out = chain.getRegressorGram(q, dq, ddq, tm, layout="element", chunk_samples=ch, workspace=ws)
        t = timeit(lambda: chain.getRegressorGram(q, dq, ddq, tm, layout="element", chunk_samples=ch, out=out, workspace=ws), reps=5, warm=2)
        print("  fused chunk=%7d (scratch %.0f MB): %.1f us -> %.3e evals/s, %.2f TFLOP/s (dense-syrk flops)" % (
            ch, ch * n * (P + 1) * 8 / 1e6, t * 1e6, N / t, flops_gram(n, P) * N / t / 1e12))


if __name__ == "__main__":
    run("config 2 chain", "ur10_like.urdf", "base_link", "wrist_3_link", 1000000, [8192, 32768, 131072, 1000000])
    run("config 3 (Panda-like, no flange)", "panda_like.urdf", "link0", "link7", 4000000, [16384, 32768, 131072])
