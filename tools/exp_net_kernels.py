"""Steady-state kernels of one Value / SLPolicy forward (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iago_amd import network
which, n = sys.argv[1], int(sys.argv[2])
m = (network.Value() if which == "value" else network.SLPolicy()).eval().cuda()
x = (torch.rand(n, 2, 8, 8, device="cuda") < 0.3).float()
with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    from torch.cuda import profiler
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        m(x)
    e1.record()
    torch.cuda.synchronize()
print(which, n, "%.3f ms per forward" % (e0.elapsed_time(e1) / 200))
