"""Split-f16 convolution: time vs input channels (stage cost vs fixed cost) and batch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iago_amd import ops

def t(n, cin, reps=50):
    a = ops.split_nchw(torch.rand(n, cin, 8, 8, device="cuda"))
    w_hi, w_lo = ops.split_weights(torch.randn(128, cin, 3, 3, device="cuda") * 0.03)
    b = torch.zeros(128, device="cuda")
    for _ in range(5):
        ops.conv3x3_split(a, w_hi, w_lo, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv3x3_split(a, w_hi, w_lo, b)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for n in (4, 256, 1024, 2048, 4096):
    r = [t(n, c) for c in (32, 64, 128, 256)]
    print("n=%5d  cin 32/64/128/256: %s us   per 3-tap stage %.2f us, fixed %.1f us" % (
        n, " ".join("%.1f" % x for x in r), (r[3] - r[2]) / 24, r[2] - (r[3] - r[2])))
