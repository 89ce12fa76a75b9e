"""Value forward: split-f16 MFMA convolutions vs MIOpen float32, per batch size."""
import sys, time
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iago_amd import network, ops

torch.manual_seed(0)
m = network.Value().eval().cuda()
for n in (64, 256, 1024, 4096):
    x = (torch.rand(n, 2, 8, 8, device="cuda") < 0.3).float()
    for mode in (False, True):
        m.split_f16 = mode
        with torch.no_grad():
            for _ in range(5):
                m(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 30
            e0.record()
            for _ in range(reps):
                m(x)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        flops = n * 122994944
        print("n=%5d split=%-5s %.3f ms  %.1f TFLOP/s-equivalent" % (n, mode, ms, flops / ms / 1e9))
# one layer alone
a = ops.split_nchw(torch.rand(1024, 128, 8, 8, device="cuda"))
w_hi, w_lo = ops.split_weights(torch.randn(128, 128, 3, 3, device="cuda") * 0.03)
b = torch.zeros(128, device="cuda")
for _ in range(3):
    ops.conv3x3_split(a, w_hi, w_lo, b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    ops.conv3x3_split(a, w_hi, w_lo, b)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
print("conv3x3_split 1024 boards 128->128: %.1f us, %.1f TFLOP/s-equivalent" % (ms * 1e3, 1024 * 64 * 128 * 1152 * 2 / ms / 1e9))
