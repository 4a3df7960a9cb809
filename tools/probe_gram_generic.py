import sys, time, torch
sys.path.insert(0, "/root/repo")
from rosdyn_amd.gram import gram
N, P = 6000000, 60
A = torch.rand((P, N), dtype=torch.float64, device="cuda")   # column-major (N x P)
b = torch.rand((N,), dtype=torch.float64, device="cuda")
out = gram(A, b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): gram(A, b, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("rdyn_gram 6e6 x 60: %.3f ms  %.0f GB/s read" % (ms, (N * (P + 1) * 8) / ms / 1e6))
