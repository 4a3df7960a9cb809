#!/usr/bin/env python3
"""GPU probes: HBM write/copy ceilings (torch fill/copy) and the regressor kernel with its Y stores
redirected to an L2-resident buffer (compute + store-issue time without HBM write-back)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain                                   # noqa: E402
from rosdyn_amd._lib import Batch, RegressorLayout, check, lib  # noqa: E402


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    dev = torch.device("cuda:0")
    N, n, P = 1000000, 6, 60
    Y = torch.empty((P, n, N), dtype=torch.float64, device=dev)
    X = torch.empty_like(Y)
    nb = Y.numel() * 8
    t = timeit(lambda: Y.zero_())
    print("fill  %.2f GB : %.1f us  %.0f GB/s (write only)" % (nb / 1e9, t * 1e6, nb / t / 1e9))
    t = timeit(lambda: Y.copy_(X))
    print("copy  %.2f GB : %.1f us  %.0f GB/s (read+write)" % (nb / 1e9, t * 1e6, 2 * nb / t / 1e9))
    chain = Chain(os.path.join(ROOT, "tests/fixtures/ur10_like.urdf"), "base_link", "wrist_3_link", (0, 0, -9.806))
    q, dq, ddq = (torch.rand((n, N), dtype=torch.float64, device=dev) * 2 - 1 for _ in range(3))
    tau = torch.empty((n, N), dtype=torch.float64, device=dev)
    b = Batch(N, q.data_ptr(), dq.data_ptr(), ddq.data_ptr(), 1, 0, torch.cuda.current_stream().cuda_stream)

    def run(yl, tau_ptr):
        check(lib().rdyn_regressor(chain._h, C.byref(b), tau_ptr, Y.data_ptr(), C.byref(yl)))

    full = RegressorLayout(1, N, n * N)
    l2 = RegressorLayout(1, 0, 0)
    t = timeit(lambda: run(full, tau.data_ptr()))
    print("regressor, Y to HBM          : %.1f us  %.0f GB/s algorithmic" % (t * 1e6, 3072 * N / t / 1e9))
    t = timeit(lambda: run(l2, tau.data_ptr()))
    print("regressor, Y to one 8 MB row : %.1f us  (compute + store issue, no HBM write-back)" % (t * 1e6))
    t = timeit(lambda: check(lib().rdyn_joint_torque(chain._h, C.byref(b), tau.data_ptr())))
    print("joint torque only            : %.1f us  %.3e evals/s  %.0f GB/s (192 B/eval)" % (t * 1e6, N / t, 192 * N / t / 1e9))
    M = torch.empty((n, n, N), dtype=torch.float64, device=dev)
    t = timeit(lambda: check(lib().rdyn_joint_inertia(chain._h, C.byref(b), M.data_ptr())))
    print("joint inertia                : %.1f us  %.3e evals/s  %.0f GB/s (336 B/eval)" % (t * 1e6, N / t, 336 * N / t / 1e9))
    T = torch.empty((7, 12, N), dtype=torch.float64, device=dev)
    t = timeit(lambda: check(lib().rdyn_transformation(chain._h, C.byref(b), None, T.data_ptr())))
    print("transformations (all links)  : %.1f us  %.3e evals/s  %.0f GB/s (720 B/eval)" % (t * 1e6, N / t, 720 * N / t / 1e9))


if __name__ == "__main__":
    main()
