"""SLPolicy / Value forward at small batches: float32 matrix-unit kernels vs MIOpen."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iago_amd import network

def t(m, x, reps=100):
    with torch.no_grad():
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            m(x)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

p = network.SLPolicy().eval().cuda()
v = network.Value().eval().cuda()
for n in (1, 8, 32, 64, 128, 256):
    x = (torch.rand(n, 2, 8, 8, device="cuda") < 0.3).float()
    p.F32_MAX_BATCH = 1 << 30
    a = t(p, x)
    p.F32_MAX_BATCH = 0
    b = t(p, x)
    v.split_f16, v.SPLIT_MIN_BATCH = True, 1 << 30
    c = t(v, x)
    v.SPLIT_MIN_BATCH = 1
    c2 = t(v, x)
    v.split_f16 = False
    d = t(v, x)
    print("n=%4d  policy: kernels %.0f us, MIOpen %.0f us   value: f32 kernels %.0f us, split-f16 %.0f us, MIOpen %.0f us" % (n, a, b, c, c2, d))
