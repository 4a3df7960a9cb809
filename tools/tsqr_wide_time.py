#!/usr/bin/env python3
"""rdyn_tsqr on materialised matrices of 97..112 columns: time per call and the kernels behind it."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd.gram import tsqr             # noqa: E402
from tools.probe import timeit               # noqa: E402

for rws, cols in ((1000000, 111), (1000000, 96), (1000000, 95), (4000000, 111), (6000000, 85)):
    Am = torch.rand((cols, rws), dtype=torch.float64, device="cuda")
    bm = torch.rand((rws,), dtype=torch.float64, device="cuda")
    t = timeit(lambda: tsqr(Am, bm), reps=3, warm=1)
    print("rdyn_tsqr: %8d rows x (%3d + 1) columns from memory   %9.1f us   %6.0f GB/s of reads" % (rws, cols, t * 1e6, 8 * (cols + 1) * rws / t / 1e9), flush=True)
    del Am, bm
