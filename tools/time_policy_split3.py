#!/usr/bin/env python3
"""Time of the SLPolicy forward on n boards: per-layer float32 kernels (stem_f32_boards + 7 x
conv3x3_f32 + policy_head) against the one-launch three-piece split (iago_policy_forward_split3)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from iago_amd import network, ops  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
gold = np.load(os.path.join(GOLDEN, "nets_shipped.npz"))
m = network.SLPolicy().load_npz(os.path.join(GOLDEN, "sl_model.npz")).cuda().eval()
for n in (1, 2, 4, 8, 16, 32, 64, 128, 170, 192, 256, 512):
    idx = np.arange(n) % 256
    own, opp = ops.bits_to_tensor(gold["own"][idx]), ops.bits_to_tensor(gold["opp"][idx])
    index = torch.arange(n, device="cuda")
    n_dev = torch.tensor([n], dtype=torch.int32, device="cuda")
    res = {}
    for name in ("f32", "split3"):
        m.split3 = name == "split3"
        if name == "f32" and n > 192:
            continue
        for _ in range(5):
            p = m.forward_counted_boards(own, opp, index, n, n_dev)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            p = m.forward_counted_boards(own, opp, index, n, n_dev)
        e1.record()
        torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) / 50 * 1e3, p)
    line = "n %4d " % n + " ".join("%s %7.1f us" % (k, v[0]) for k, v in res.items())
    if len(res) == 2:
        line += "  max diff %.2e" % (res["f32"][1][:n] - res["split3"][1][:n]).abs().max().item()
    print(line, flush=True)
