#!/usr/bin/env python3
"""Standalone duration of the device-counted Value forward (one board per workgroup, persistent
grid) at n rows, alone and sharing its launch with a 1024-board rollout (value_rollout_kernel)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from iago_amd import network, ops  # noqa: E402

torch.manual_seed(0)
m = network.Value().cuda().eval()
w, b = bench.shipped_rollout_weights()
W = ops.RolloutWeights(w, b)
g = torch.Generator().manual_seed(1)
N = 1024
own = torch.randint(0, 2 ** 62, (N,), generator=g).cuda()
opp = torch.randint(0, 2 ** 62, (N,), generator=g).cuda() & ~own
out = torch.zeros(N, device="cuda")
index = torch.randperm(N, device="cuda")


def timed(fn, reps=20):
    """fn captured `reps` times in one hipGraph (no host time between the launches)."""
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        s.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


for n in (0, 1, 32, 64, 128, 165, 200, 256, 257, 320, 512, 1024):
    n_dev = torch.tensor([n], dtype=torch.int32, device="cuda")
    with torch.no_grad():
        t = timed(lambda: m.forward_boards_counted(own, opp, index, n_dev, out))
    print("n %4d: value alone %6.1f us" % (n, t), flush=True)
