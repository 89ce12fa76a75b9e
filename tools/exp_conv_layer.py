"""One 128->128 layer of the split-f16 convolution, 1024 boards, repeated (for rocprofv3)."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iago_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
a = ops.split_nchw(torch.rand(n, 128, 8, 8, device="cuda"))
w_hi, w_lo = ops.split_weights(torch.randn(128, 128, 3, 3, device="cuda") * 0.03)
b = torch.zeros(128, device="cuda")
for _ in range(20):
    ops.conv3x3_split(a, w_hi, w_lo, b)
torch.cuda.synchronize()
