#!/usr/bin/env python3
"""Time of one Value forward (split-f16 stack) from bitboards: python3 tools/time_value.py [boards] [reps]"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iago_amd import network  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
torch.manual_seed(0)
m = network.Value().cuda().eval()
m.fused = os.environ.get("IAGO_VALUE_UNFUSED") is None   # one launch / stem + trunk + head
g = torch.Generator().manual_seed(1)
own = torch.randint(0, 2 ** 62, (n,), generator=g).cuda()
opp = torch.randint(0, 2 ** 62, (n,), generator=g).cuda() & ~own
with torch.no_grad():
    for _ in range(5):
        m.forward_boards(own, opp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        v = m.forward_boards(own, opp)
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(("fused" if m.fused else "3 launches") + " Value forward, %d boards: %.1f us; trunk MFMA flops 3 x 122.68 MFLOP x n / t = %.0f TFLOP/s executed"
      % (n, ms * 1e3, 3 * 122.683392e6 * n / (ms * 1e-3) / 1e12))
