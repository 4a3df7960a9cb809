#!/usr/bin/env python3
"""GPU probe: leading dimension of the stacked (N n) x P regressor INSIDE ONE allocation (fixed physical backing): does a padded column
stride lift a 'slow' placement to the fast rate?  (H2: the column stride aliases with the channel / bank hash of physically contiguous memory.)"""
import ctypes as C, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain
from rosdyn_amd._lib import Batch, RegressorLayout, check, lib
N = 1000000
c = Chain(os.path.join(ROOT, "tests/fixtures/ur10_like.urdf"), "base_link", "wrist_3_link", (0, 0, -9.806))
n, P = 6, 60
q, dq, ddq = (torch.rand((N, n), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(3))
pads = [0, 16, 32, 48, 64, 128, 256, 272, 512, 1024, 2048, 4096, 4112, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576]
for rep in range(3):
    big = torch.empty((P * (N * n + max(pads)),), dtype=torch.float64, device="cuda")
    out = []
    for pad in pads:
        lda = N * n + pad
        yl = RegressorLayout(n, 1, lda)
        b = Batch(); b.n_samples, b.q, b.dq, b.ddq, b.layout, b.device = N, q.data_ptr(), dq.data_ptr(), ddq.data_ptr(), 0, 0
        b.stream = torch.cuda.current_stream().cuda_stream
        f = lambda: check(lib().rdyn_regressor(c._h, C.byref(b), None, big.data_ptr(), C.byref(yl)))
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6): f()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 6 * 1e6)
    print("alloc %d @0x%x:" % (rep, big.data_ptr()), " ".join("%d:%.0f" % (p, t) for p, t in zip(pads, out)))
    keep = big  # keep alive so that the next one lands elsewhere
    if rep == 0: k0 = big
    if rep == 1: k1 = big
